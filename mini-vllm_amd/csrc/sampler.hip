// K8 — fused temperature / top-k / top-p / min-p / multinomial sampler
// (reference: minivllm/sampling/sampler.py:59-110,
//  sampling/functional.py:34-233,410-437, driven by
//  engine/inference_executor.py:776-806).
//
// One 1024-thread workgroup per logits row; the row is only ever READ
// (no sort, no [B,V] temporaries).  All three filters keep a "top set by
// value", so each one reduces to a threshold on the order-preserving
// integer key of the raw logit, found by radix select over LDS histograms:
//   top-k : k-th largest key (count histogram) + how many ties of it to keep
//   top-p : smallest key v with mass(keys > v) <= p * mass(top-k set)
//   min-p : weight threshold  exp((x-max)/T) >= min_p
// then a single multinomial draw (Philox4x32-10) over the surviving weights.
// Math is fp32 throughout (the reference computes in the logits dtype; see
// DESIGN.md "sampler parity").
#include "common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;

struct SampleParams {
  const void *logits;
  int64_t ld;
  const float *temperatures;
  const int64_t *top_ks;
  const float *top_ps;
  const float *min_ps;
  uint64_t seed, offset;
  const uint64_t *offset_dev;
  int64_t *out_tokens;
  uint8_t *keep_mask;
  float *stats;   // optional [batch][2]: {row maximum, kept mass relative to it at the row's temperature}
  int vocab;
};

constexpr int kWin = 2048;  // window histogram: the 2048 distinct 16-bit values just below the row maximum

struct Shared {
  unsigned wcnt[kWin];
  float fhist[256];
  unsigned ihist[256];
  float fred[kWaves];
  unsigned ured[kWaves];
  int ired[kWaves];
  float fscan[kWaves];
  unsigned uscan[kWaves];
  unsigned bcast_u[4];
  float bcast_f[4];
};

// ---- Philox4x32-10 -----------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], const uint32_t (&k)[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint32_t row) {
  uint32_t c[4] = {(uint32_t)offset, (uint32_t)(offset >> 32), row, 0u};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k);
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
  }
  return (float)(c[0] >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}

// a second, independent uniform of the same (seed, offset, row) counter: word 1 of the Philox output
__device__ __forceinline__ float philox_uniform_b(uint64_t seed, uint64_t offset, uint32_t row) {
  uint32_t c[4] = {(uint32_t)offset, (uint32_t)(offset >> 32), row, 0u};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k);
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
  }
  return (float)(c[1] >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}

// ---- order-preserving keys -----------------------------------------------------
template <typename Tag> struct keyinfo;
template <> struct keyinfo<bf16_tag> { static constexpr int BITS = 16; };
template <> struct keyinfo<f16_tag> { static constexpr int BITS = 16; };
template <> struct keyinfo<f32_tag> { static constexpr int BITS = 32; };

template <typename Tag>
__device__ __forceinline__ uint32_t make_key(typename elem<Tag>::T raw) {
  if constexpr (keyinfo<Tag>::BITS == 16) {
    const uint32_t u = raw;
    return (u & 0x8000u) ? (~u & 0xffffu) : (u | 0x8000u);
  } else {
    const uint32_t u = __builtin_bit_cast(uint32_t, raw);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  }
}

// ---- block-wide helpers (all 1024 threads must call) ------------------------------
__device__ __forceinline__ float block_max(float x, Shared &sh) {
  x = wave_max(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh.fred[threadIdx.x >> 6] = x;
  __syncthreads();
  float r = sh.fred[0];
#pragma unroll
  for (int i = 1; i < kWaves; ++i) r = fmaxf(r, sh.fred[i]);
  return r;
}
__device__ __forceinline__ float block_sum_f(float x, Shared &sh) {
  x = wave_sum(x);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh.fred[threadIdx.x >> 6] = x;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < kWaves; ++i) r += sh.fred[i];
  return r;
}
// exclusive prefix over threads (thread order) + block total
__device__ __forceinline__ float block_exscan_f(float x, float &total, Shared &sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float inc = x;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float y = __shfl_up(inc, off, 64);
    if (lane >= off) inc += y;
  }
  __syncthreads();
  if (lane == 63) sh.fscan[w] = inc;
  __syncthreads();
  float base = 0.f, tot = 0.f;
#pragma unroll
  for (int i = 0; i < kWaves; ++i) {
    if (i < w) base += sh.fscan[i];
    tot += sh.fscan[i];
  }
  total = tot;
  return base + inc - x;
}
__device__ __forceinline__ unsigned block_exscan_u(unsigned x, unsigned &total, Shared &sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned inc = x;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned y = __shfl_up(inc, off, 64);
    if (lane >= off) inc += y;
  }
  __syncthreads();
  if (lane == 63) sh.uscan[w] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kWaves; ++i) {
    if (i < w) base += sh.uscan[i];
    tot += sh.uscan[i];
  }
  total = tot;
  return base + inc - x;
}

// Row walker: thread t owns 16-byte vectors t, t+1024, ... (or single elements
// when the row is not 16-byte aligned).  Every pass visits a thread's elements
// in the same order, which is what the tie ranks and the final draw rely on.
// All threads run the same trip count and call f(index, raw, in_range) with
// the wave fully converged (f may use cross-lane operations: DPP / bpermute
// read stale registers from lanes that left a divergent loop).
template <typename Tag, typename F>
__device__ __forceinline__ void row_foreach(const typename elem<Tag>::T *row, int vocab, bool vec_ok, F &&f) {
  typedef typename elem<Tag>::T T;
  constexpr int VEC = elem<Tag>::VEC;
  int scalar_from = 0;
  if (vec_ok) {
    const int nvec = vocab / VEC;
    constexpr int U = 4;  // loads in flight per thread: a pass is L2/MALL-latency bound otherwise
    for (int c0 = 0; c0 < nvec; c0 += kThreads * U) {
      u32x4 raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * kThreads + threadIdx.x;
        raw[u] = u32x4{0u, 0u, 0u, 0u};
        if (c < nvec) raw[u] = *reinterpret_cast<const u32x4 *>(row + (int64_t)c * VEC);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * kThreads + threadIdx.x;
        const bool ok = c < nvec;
        if (c0 + u * kThreads >= nvec) break;  // block-uniform: nobody has work in this sub-batch
        if constexpr (VEC == 8) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f(c * VEC + 2 * j, (T)(raw[u][j] & 0xffffu), ok);
            f(c * VEC + 2 * j + 1, (T)(raw[u][j] >> 16), ok);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) f(c * VEC + j, u2f(raw[u][j]), ok);
        }
      }
    }
    scalar_from = nvec * VEC;
  }
  for (int i0 = scalar_from; i0 < vocab; i0 += kThreads) {
    const int i = i0 + threadIdx.x;
    const bool ok = i < vocab;
    T raw = (T)0;
    if (ok) raw = row[i];
    f(i, raw, ok);
  }
}

// LDS histogram add: a plain LDS atomic.  (Rounds 1-2 took the wave's most common bins out first with a ballot /
// broadcast / wave-sum "leader" step; measured on gfx950 that cost more than the same-address serialisation it avoided:
// the radix paths ran 1.1-3.1x faster without it, profiles/r03_nucleus_sequence_us.json.)
template <typename V>
__device__ __forceinline__ void hist_add(V *hist, bool active, unsigned bin, V val) {
  if (active) atomicAdd(&hist[bin], val);
}

template <typename Tag>
__global__ void __launch_bounds__(kThreads) sample_kernel(SampleParams p) {
  typedef typename elem<Tag>::T T;
  constexpr int KBITS = keyinfo<Tag>::BITS;
  constexpr int LEVELS = KBITS / 8;
  __shared__ Shared sh;

  const int rowi = blockIdx.x;
  const int V = p.vocab;
  const T *row = (const T *)p.logits + (int64_t)rowi * p.ld;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
  uint8_t *mask_row = p.keep_mask ? p.keep_mask + (int64_t)rowi * V : nullptr;
  const float temp = p.temperatures ? p.temperatures[rowi] : 1.0f;

  // ---- pass A: max (and argmax, lowest index on ties) ------------------------------
  float best = -INFINITY;
  int best_i = 0x7fffffff;
  row_foreach<Tag>(row, V, vec_ok, [&](int i, T raw, bool ok) {
    const float x = elem<Tag>::ld(raw);
    if (ok && (x > best || (x == best && i < best_i))) { best = x; best_i = i; }
  });
  const float xmax = block_max(best, sh);
  if (temp == 0.0f) {  // greedy (argmax, not multinomial over softmax(l/1e-8))
    int cand = (best == xmax) ? best_i : 0x7fffffff;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cand = min(cand, __shfl_xor(cand, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.ired[threadIdx.x >> 6] = cand;
    __syncthreads();
    int win = sh.ired[0];
#pragma unroll
    for (int i = 1; i < kWaves; ++i) win = min(win, sh.ired[i]);
    if (win == 0x7fffffff) win = 0;  // all-NaN row
    if (threadIdx.x == 0) p.out_tokens[rowi] = win;
    if (threadIdx.x == 0 && p.stats) {
      p.stats[2 * rowi] = xmax;
      p.stats[2 * rowi + 1] = 1.0f;
    }
    if (mask_row)
      for (int i = threadIdx.x; i < V; i += kThreads) mask_row[i] = (i == win);
    return;
  }

  const bool degenerate = !(xmax > -INFINITY) || !(xmax < INFINITY);  // all -inf / NaN / +inf
  const float inv_t = 1.4426950408889634f / fmaxf(temp, 1e-8f);
  auto weight = [&](float x) -> float {
    return degenerate ? 1.0f : __builtin_amdgcn_exp2f((x - xmax) * inv_t);
  };

  // ---- top-k: k-th largest key, and how many of its ties to keep ----------------------
  const int64_t k_raw = p.top_ks ? p.top_ks[rowi] : -1;
  const bool topk_on = !degenerate && k_raw > 0 && k_raw < (int64_t)V;
  const float top_p = p.top_ps ? p.top_ps[rowi] : 1.0f;
  const bool topp_on = !degenerate && top_p < 1.0f;
  uint32_t key_k = 0;        // keep key > key_k, plus `need_ties` elements with key == key_k
  unsigned need_ties = 0;
  bool ties_partial = false; // true when only some of the ties survive
  unsigned tie_before = 0;   // this thread's first tie rank (thread-major order)
  uint32_t key_p = 0;        // keep key >= key_p
  bool topk_done = false, topp_done = false;

  // ---- 16-bit logits: ONE histogram pass resolves both thresholds.  Every bin of the window is
  //      exactly one representable value (bin = key(max) - key), so there is no second radix level;
  //      values below the window only feed a count and a mass.  If a threshold falls below the
  //      window (flat distributions, huge k) the generic radix select further down takes over.
  if constexpr (KBITS == 16) {
    if (topk_on || topp_on) {
      const uint32_t kmax = make_key<Tag>(elem<Tag>::st(xmax));
      for (int i = threadIdx.x; i < kWin; i += kThreads) sh.wcnt[i] = 0u;
      __syncthreads();
      // only COUNTS are accumulated: all members of a bin are the same value, so its mass is
      // count * weight(value).  Values below the window need their mass only when top-k is off.
      unsigned under_c = 0;
      float under_m = 0.f;
      row_foreach<Tag>(row, V, vec_ok, [&](int i, T raw, bool ok) {
        const uint32_t bin = kmax - make_key<Tag>(raw);
        if (ok) {
          if (bin < (uint32_t)kWin) {
            atomicAdd(&sh.wcnt[bin], 1u);
          } else {
            ++under_c;
            if (!topk_on) under_m += weight(elem<Tag>::ld(raw));
          }
        }
      });
      __syncthreads();
      auto bin_weight = [&](uint32_t bin) -> float {  // weight of the value that owns `bin`
        const uint32_t key = kmax - bin;              // invert make_key: sortable key -> raw bits
        const uint32_t rawbits = (key & 0x8000u) ? (key & 0x7fffu) : (~key & 0xffffu);
        return weight(elem<Tag>::ld((T)rawbits));
      };
      const unsigned c0 = sh.wcnt[2 * threadIdx.x], c1 = sh.wcnt[2 * threadIdx.x + 1];
      const float m0 = c0 ? (float)c0 * bin_weight(2 * threadIdx.x) : 0.f;
      const float m1 = c1 ? (float)c1 * bin_weight(2 * threadIdx.x + 1) : 0.f;
      unsigned win_c;
      float win_m;
      const unsigned pc = block_exscan_u(c0 + c1, win_c, sh);   // elements in bins before mine
      const float pm = block_exscan_f(m0 + m1, win_m, sh);      // their mass
      const float under_mass = block_sum_f(under_m, sh);
      unsigned under_cnt;
      (void)block_exscan_u(under_c, under_cnt, sh);
      __syncthreads();
      if (threadIdx.x == 0) { sh.bcast_u[0] = 0xffffffffu; sh.bcast_u[3] = 0u; }
      __syncthreads();
      int limit = kWin - 1;       // last bin that still has top-k members
      float z1 = win_m + under_mass;
      bool ok_fast = true;
      if (topk_on) {
        if (win_c >= (unsigned)k_raw) {
          const unsigned kk = (unsigned)k_raw;
          if (pc < kk && kk <= pc + c0 + c1) {  // exactly one thread
            const bool second = pc + c0 < kk;
            const unsigned above = second ? pc + c0 : pc;
            const unsigned bc = second ? c1 : c0;
            const float bm = second ? m1 : m0;
            const unsigned need = kk - above;
            sh.bcast_u[0] = 2 * threadIdx.x + (second ? 1 : 0);
            sh.bcast_u[1] = need;
            sh.bcast_u[2] = bc;
            sh.bcast_f[1] = (second ? pm + m0 : pm) + (float)need * (bm / (float)bc);  // mass of the top-k set
          }
          __syncthreads();
          const unsigned bk = sh.bcast_u[0];
          key_k = kmax - bk;
          need_ties = sh.bcast_u[1];
          ties_partial = need_ties < sh.bcast_u[2];
          z1 = sh.bcast_f[1];
          limit = (int)bk;
          topk_done = true;
        } else {
          ok_fast = false;  // the k-th value lies below the window
        }
      }
      if (ok_fast && topp_on) {
        // largest bin index i <= limit whose strictly-larger mass E_i fits the budget (E_0 = 0 always does)
        const float budget = top_p * z1;
        unsigned cand = 0;
        const int b0 = 2 * threadIdx.x, b1 = b0 + 1;
        if (b0 <= limit && pm <= budget) cand = (unsigned)b0 + 1u;
        if (b1 <= limit && pm + m0 <= budget) cand = (unsigned)b1 + 1u;
        if (cand) atomicMax(&sh.bcast_u[3], cand);
        __syncthreads();
        const unsigned bp = sh.bcast_u[3] - 1u;  // >= 0: bin 0 always qualifies
        // without top-k, "the whole window fits" means the nucleus edge is below the window
        if (!topk_on && under_cnt > 0 && bp == (unsigned)(kWin - 1) && win_m <= budget) {
          ok_fast = false;
        } else {
          key_p = kmax - bp;
          topp_done = true;
        }
      }
      if (ties_partial && topk_done) {
        unsigned mine_t = 0;
        row_foreach<Tag>(row, V, vec_ok, [&](int i, T raw, bool ok) { mine_t += ok && make_key<Tag>(raw) == key_k; });
        unsigned tot;
        tie_before = block_exscan_u(mine_t, tot, sh);
      }
    }
  }

  if (topk_on && !topk_done) {
    uint32_t prefix = 0;
    unsigned above = 0;      // elements with a strictly larger prefix
    unsigned bin_count = 0;
    for (int lvl = 0; lvl < LEVELS; ++lvl) {
      const int shift = KBITS - 8 * (lvl + 1);
      __syncthreads();
      if (threadIdx.x < 256) sh.ihist[threadIdx.x] = 0;
      __syncthreads();
      row_foreach<Tag>(row, V, vec_ok, [&](int i, T raw, bool ok) {
        const uint32_t key = make_key<Tag>(raw);
        const bool in = ok && (lvl == 0 || (key >> (shift + 8)) == prefix);
        hist_add<unsigned>(sh.ihist, in, (key >> shift) & 0xffu, 1u);
      });
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned c = above;
        int d = 255;
        for (; d > 0; --d) {
          if (c + sh.ihist[d] >= (unsigned)k_raw) break;
          c += sh.ihist[d];
        }
        sh.bcast_u[0] = (unsigned)d;
        sh.bcast_u[1] = c;
        sh.bcast_u[2] = sh.ihist[d];
      }
      __syncthreads();
      prefix = (prefix << 8) | sh.bcast_u[0];
      above = sh.bcast_u[1];
      bin_count = sh.bcast_u[2];
    }
    key_k = prefix;
    need_ties = (unsigned)k_raw - above;  // 1 <= need_ties <= bin_count
    ties_partial = need_ties < bin_count;
    if (ties_partial) {
      unsigned mine = 0;
      row_foreach<Tag>(row, V, vec_ok, [&](int i, T raw, bool ok) { mine += ok && make_key<Tag>(raw) == key_k; });
      unsigned tot;
      tie_before = block_exscan_u(mine, tot, sh);
    }
  }
  // membership in the top-k set; `tie_seen` must be reset to 0 before every pass
  unsigned tie_seen = 0;
  auto in_topk = [&](uint32_t key) -> bool {
    if (!topk_on) return true;
    if (key > key_k) return true;
    if (key < key_k) return false;
    if (!ties_partial) return true;
    return (tie_before + tie_seen++) < need_ties;
  };

  // ---- top-p: smallest key v with mass(key > v) <= p * Z1 -----------------------------
  if (topp_on && !topp_done) {
    float z = 0.f;
    tie_seen = 0;
    row_foreach<Tag>(row, V, vec_ok, [&](int i, T raw, bool ok) {
      if (ok && in_topk(make_key<Tag>(raw))) z += weight(elem<Tag>::ld(raw));
    });
    const float budget = top_p * block_sum_f(z, sh);
    uint32_t prefix = 0;
    float above = 0.f;
    for (int lvl = 0; lvl < LEVELS; ++lvl) {
      const int shift = KBITS - 8 * (lvl + 1);
      __syncthreads();
      if (threadIdx.x < 256) sh.fhist[threadIdx.x] = 0.f;
      __syncthreads();
      tie_seen = 0;
      row_foreach<Tag>(row, V, vec_ok, [&](int i, T raw, bool ok) {
        const uint32_t key = make_key<Tag>(raw);
        const bool member = ok && in_topk(key);
        const bool in = member && (lvl == 0 || (key >> (shift + 8)) == prefix);
        hist_add<float>(sh.fhist, in, (key >> shift) & 0xffu, in ? weight(elem<Tag>::ld(raw)) : 0.f);
      });
      __syncthreads();
      if (threadIdx.x == 0) {
        // walk down while everything strictly above the NEXT bin still fits the budget
        float c = above;
        int d = 255;
        for (; d > 0; --d) {
          const float next = c + sh.fhist[d];
          if (!(next <= budget)) break;
          c = next;
        }
        sh.bcast_u[0] = (unsigned)d;
        sh.bcast_f[0] = c;
      }
      __syncthreads();
      prefix = (prefix << 8) | sh.bcast_u[0];
      above = sh.bcast_f[0];
    }
    key_p = prefix;
  }

  // ---- min-p: prob < min_p * max_prob  <=>  weight < min_p  (the normaliser cancels) ----
  const float min_p = p.min_ps ? p.min_ps[rowi] : 0.0f;
  const bool minp_on = !degenerate && min_p > 0.0f;

  auto kept = [&](uint32_t key, float w) -> bool {
    bool k = in_topk(key);
    if (topp_on) k = k && key >= key_p;
    if (minp_on) k = k && !(w < min_p);
    return k;
  };

  // ---- final: per-thread mass, exclusive scan in thread order, one draw -----------------
  float mine = 0.f;
  tie_seen = 0;
  row_foreach<Tag>(row, V, vec_ok, [&](int i, T raw, bool ok) {
    const float w = weight(elem<Tag>::ld(raw));
    const bool k = ok && kept(make_key<Tag>(raw), w);
    if (ok && mask_row) mask_row[i] = k;
    if (k) mine += w;
  });
  float total;
  const float before = block_exscan_f(mine, total, sh);
  const uint64_t off = p.offset + (p.offset_dev ? *p.offset_dev : 0ull);
  const float target = philox_uniform(p.seed, off, (uint32_t)rowi) * total;
  if (threadIdx.x == 0 && p.stats) {
    p.stats[2 * rowi] = xmax;
    p.stats[2 * rowi + 1] = total;
  }

  // owner = last thread whose prefix <= target and that holds mass
  __syncthreads();
  if (threadIdx.x == 0) sh.bcast_u[3] = 0u;
  __syncthreads();
  const bool owner = mine > 0.f && before <= target;
  if (owner) atomicMax(&sh.bcast_u[3], (unsigned)threadIdx.x + 1u);
  __syncthreads();
  const unsigned who = sh.bcast_u[3] ? sh.bcast_u[3] - 1u : 0u;
  // The owner's elements are re-examined by ITS WHOLE WAVE: lane L takes the owner's L-th vector
  // (or L-th scalar-tail element), so the final search is one vector per lane instead of a serial walk.
  if ((threadIdx.x >> 6) == (who >> 6)) {
    constexpr int VEC = elem<Tag>::VEC;
    const int lane = threadIdx.x & 63;
    const float before_o = __shfl(before, (int)(who & 63), 64);
    const unsigned tie_before_o = __shfl((int)tie_before, (int)(who & 63), 64);
    const int nvec = vec_ok ? V / VEC : 0;
    const int n_own_vec = nvec > (int)who ? (nvec - (int)who + kThreads - 1) / kThreads : 0;
    const int tail0 = nvec * VEC;
    const int n_own_tail = V - tail0 > (int)who ? (V - tail0 - (int)who + kThreads - 1) / kThreads : 0;
    int pick = -1, last_kept = -1;
    float run_base = before_o;
    unsigned ties_base = tie_before_o;
    for (int base = 0; base < n_own_vec + n_own_tail; base += 64) {   // wave-uniform trip count
      const int slot = base + lane;
      // this lane's elements: one vector of the owner, or one tail element
      float w[VEC];
      uint32_t key[VEC];
      int idx[VEC];
      int cnt = 0;
      if (slot < n_own_vec) {
        const int c = (int)who + slot * kThreads;
        const u32x4 raw = *reinterpret_cast<const u32x4 *>(row + (int64_t)c * VEC);
        cnt = VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          T e;
          if constexpr (VEC == 8) e = (T)((j & 1) ? (raw[j >> 1] >> 16) : (raw[j >> 1] & 0xffffu));
          else e = u2f(raw[j]);
          key[j] = make_key<Tag>(e);
          w[j] = weight(elem<Tag>::ld(e));
          idx[j] = c * VEC + j;
        }
      } else if (slot < n_own_vec + n_own_tail) {
        const int i = tail0 + (int)who + (slot - n_own_vec) * kThreads;
        const T e = row[i];
        cnt = 1;
        key[0] = make_key<Tag>(e);
        w[0] = weight(elem<Tag>::ld(e));
        idx[0] = i;
      }
      // tie ranks continue across lanes in the owner's visiting order
      unsigned my_ties = 0;
#pragma unroll
      for (int j = 0; j < VEC; ++j) my_ties += (j < cnt && topk_on && ties_partial && key[j] == key_k);
      unsigned tie_inc = my_ties;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned y = __shfl_up(tie_inc, off, 64);
        if (lane >= off) tie_inc += y;
      }
      unsigned tie_rank = ties_base + tie_inc - my_ties;
      float mass = 0.f;
      bool keepj[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        bool k = j < cnt;
        if (k && topk_on) {
          if (key[j] < key_k) k = false;
          else if (key[j] == key_k && ties_partial) k = (tie_rank++) < need_ties;
        }
        if (k && topp_on) k = key[j] >= key_p;
        if (k && minp_on) k = !(w[j] < min_p);
        keepj[j] = k;
        if (k) mass += w[j];
      }
      float inc = mass;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const float y = __shfl_up(inc, off, 64);
        if (lane >= off) inc += y;
      }
      float run = run_base + inc - mass;   // mass of everything before this lane's elements
      int lp = -1, lk = -1;
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        if (keepj[j]) {
          lk = idx[j];
          run += w[j];
          if (lp < 0 && run > target) lp = idx[j];
        }
      // first lane (in visiting order) that crossed the target wins; remember the last kept element
      const unsigned long long hit = __ballot(lp >= 0);
      if (pick < 0 && hit) pick = __shfl(lp, __ffsll((long long)hit) - 1, 64);
      const unsigned long long anyk = __ballot(lk >= 0);
      if (anyk) last_kept = __shfl(lk, 63 - __clzll((long long)anyk), 64);
      run_base += __shfl(inc, 63, 64);
      ties_base += __shfl((int)tie_inc, 63, 64);
      if (pick >= 0) break;  // wave-uniform
    }
    if (pick < 0) pick = last_kept >= 0 ? last_kept : best_i;  // rounding at the very end
    if (pick < 0 || pick >= V) pick = 0;
    if (threadIdx.x == who) p.out_tokens[rowi] = pick;
  }
}


// f2 for SAMPLED rows under tensor parallelism: instead of all-gathering [B, V/tp] logits, every
// rank reduces its vocab shard of a row to its kTopC largest logits (+ their global indices).  For a row whose
// top_k is in 1..kTopC (or that is greedy) the union of the per-rank candidates contains a complete top-k set, the
// filters after it (top-p, min-p) and the draw only ever look at that set, so running the ordinary sampler on the
// gathered [B, world * kTopC] candidates is the exact distribution of the full-row sampler (ties at the k-th value
// may be resolved towards other, equal-valued tokens).  The exchange format is the greedy path's: 16-bit floats that
// are exact (the logit itself, three 8-bit pieces of the global index) in slot [rank] of a zeroed buffer, summed by
// the peer-to-peer all-reduce (x + 0 = x): no backend collective, so the decode step stays graph-capturable.
constexpr int kTopC = 64;

template <typename Tag>
__global__ void __launch_bounds__(kThreads) topc_pack_kernel(const uint16_t *logits, int64_t ld, int vocab_local, int rank,
                                                             int world, int64_t vocab_per_rank, int rows_padded,
                                                             uint16_t *buf) {
  typedef uint16_t T;
  __shared__ Shared sh;
  const int rowi = blockIdx.x;
  const T *row = logits + (int64_t)rowi * ld;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
  // my slots <- (-inf, 0, 0, 0), every other rank's slots of this row <- 0
  const uint32_t ninf = elem<Tag>::st(-INFINITY);
  for (int i = threadIdx.x; i < world * kTopC; i += kThreads) {
    const int q = i / kTopC, c = i - q * kTopC;
    u32x2 w = {0u, 0u};
    if (q == rank) w[0] = ninf;
    *(u32x2 *)(buf + (((int64_t)q * rows_padded + rowi) * kTopC + c) * 4) = w;
  }
  // the kTopC-th largest key of the shard: two 8-bit radix levels over LDS count histograms
  const unsigned want = (unsigned)min(kTopC, vocab_local);
  uint32_t prefix = 0;
  unsigned above = 0, bin_count = 0;
  for (int lvl = 0; lvl < 2; ++lvl) {
    const int shift = 16 - 8 * (lvl + 1);
    __syncthreads();
    if (threadIdx.x < 256) sh.ihist[threadIdx.x] = 0;
    __syncthreads();
    row_foreach<Tag>(row, vocab_local, vec_ok, [&](int i, T raw, bool ok) {
      const uint32_t key = make_key<Tag>(raw);
      const bool in = ok && (lvl == 0 || (key >> (shift + 8)) == prefix);
      hist_add<unsigned>(sh.ihist, in, (key >> shift) & 0xffu, 1u);
    });
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned c = above;
      int d = 255;
      for (; d > 0; --d) {
        if (c + sh.ihist[d] >= want) break;
        c += sh.ihist[d];
      }
      sh.bcast_u[0] = (unsigned)d;
      sh.bcast_u[1] = c;
      sh.bcast_u[2] = sh.ihist[d];
    }
    __syncthreads();
    prefix = (prefix << 8) | sh.bcast_u[0];
    above = sh.bcast_u[1];
    bin_count = sh.bcast_u[2];
  }
  const uint32_t key_k = prefix;
  const unsigned need_ties = want - above;   // of the bin_count elements equal to the threshold
  (void)bin_count;
  // deterministic compaction in visiting order: count, scan, write
  unsigned my_gt = 0, my_tie = 0;
  row_foreach<Tag>(row, vocab_local, vec_ok, [&](int i, T raw, bool ok) {
    const uint32_t key = make_key<Tag>(raw);
    my_gt += ok && key > key_k;
    my_tie += ok && key == key_k;
  });
  unsigned tot;
  unsigned gt_at = block_exscan_u(my_gt, tot, sh);
  unsigned tie_at = block_exscan_u(my_tie, tot, sh);
  uint16_t *mine = buf + ((int64_t)rank * rows_padded + rowi) * kTopC * 4;
  __syncthreads();   // the fill above is complete before the candidates overwrite it
  row_foreach<Tag>(row, vocab_local, vec_ok, [&](int i, T raw, bool ok) {
    if (!ok) return;
    const uint32_t key = make_key<Tag>(raw);
    int slot = -1;
    if (key > key_k) slot = (int)gt_at++;
    else if (key == key_k) {
      const unsigned t = tie_at++;
      if (t < need_ties) slot = (int)(above + t);
    }
    if (slot >= 0 && slot < kTopC) {
      const int64_t g = (int64_t)rank * vocab_per_rank + i;
      u32x2 w;
      w[0] = (uint32_t)raw | ((uint32_t)elem<Tag>::st((float)((g >> 16) & 255)) << 16);
      w[1] = (uint32_t)elem<Tag>::st((float)((g >> 8) & 255)) | ((uint32_t)elem<Tag>::st((float)(g & 255)) << 16);
      *(u32x2 *)(mine + (int64_t)slot * 4) = w;
    }
  });
}

template <typename Tag>
__global__ void topc_unpack_kernel(const uint16_t *buf, int world, int rows, int rows_padded, uint16_t *cand_logits,
                                   int32_t *cand_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over rows * world * kTopC
  const int per_row = world * kTopC;
  if (i >= rows * per_row) return;
  const int r = i / per_row, j = i - r * per_row, q = j / kTopC, c = j - q * kTopC;
  const u32x2 w = *(const u32x2 *)(buf + (((int64_t)q * rows_padded + r) * kTopC + c) * 4);
  cand_logits[i] = (uint16_t)(w[0] & 0xffffu);
  cand_idx[i] = ((int32_t)elem<Tag>::ld((uint16_t)(w[0] >> 16)) << 16) | ((int32_t)elem<Tag>::ld((uint16_t)(w[1] & 0xffffu)) << 8) |
                (int32_t)elem<Tag>::ld((uint16_t)(w[1] >> 16));
}

// candidate position -> token id.  Sampled rows: the global index of the drawn candidate.  Greedy rows
// (temperature 0): the LOWEST global index among the candidates equal to the row maximum — candidates are laid out in
// shard-visiting order, not in index order, so "first maximum by candidate position" could name a higher token id than
// TP = 1 (argmax: lowest index on ties) and the all-greedy path (mv_greedy_merge) do, and the same greedy
// request would then decode differently depending on what it is batched with.  One wave per row.
template <typename Tag>
__global__ void __launch_bounds__(64) topc_pick_kernel(const uint16_t *cand_logits, const int32_t *cand_idx,
                                                       const int64_t *pos, const float *temperatures, int64_t *tokens,
                                                       int width) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const uint16_t *cl = cand_logits + (int64_t)r * width;
  const int32_t *ci = cand_idx + (int64_t)r * width;
  const bool greedy = temperatures != nullptr && temperatures[r] == 0.0f;
  if (!greedy) {
    if (lane == 0) {
      int64_t p = pos[r];
      p = p < 0 ? 0 : (p >= width ? width - 1 : p);
      tokens[r] = ci[p];
    }
    return;
  }
  uint32_t best = 0;
  for (int i = lane; i < width; i += 64) best = max(best, make_key<Tag>(cl[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, o, 64));
  int32_t tok = 0x7fffffff;
  for (int i = lane; i < width; i += 64)
    if (make_key<Tag>(cl[i]) == best) tok = min(tok, ci[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tok = min(tok, __shfl_xor(tok, o, 64));
  if (lane == 0) tokens[r] = tok;
}

// f2 for rows sampled from the WHOLE vocabulary (the default SamplingParams: top_k = -1, top_p = 1, min_p = 0)
// under tensor parallelism: an exact TWO-STAGE draw instead of the [B, V/tp] all-gather.  Every rank runs the ordinary
// sampler on its own shard (token t_r drawn with the row's uniform u, row maximum m_r, mass S_r = sum exp((l - m_r)/T)),
// the ranks exchange {m_r, S_r, t_r} per row — 16 bytes — through the slot scheme of the greedy path (exact 16-bit pieces
// in slot [rank] of a zeroed buffer, summed by the all-reduce), and every rank picks the SAME rank r* with probability
// S_r exp((m_r - M)/T) / Z from a second, independent uniform of the same Philox counter and takes t_{r*}:
// P(token j) = P(r*) * e_j / S_{r*} = exp((l_j - M)/T) / Z, the softmax of the full row.  Greedy rows (T = 0) take the
// rank with the largest m_r, lowest rank on ties (= the lowest token id: shards are index-ordered and a shard's argmax
// is its lowest index).  S_r travels as {binary exponent, three 8-bit pieces of a 24-bit mantissa}: integers <= 255,
// exact in bf16 and f16 alike.
template <typename Tag>
__global__ void mass_pack_kernel(const float *stats, const int64_t *local_tokens, int rank, int world,
                                 int64_t vocab_per_rank, int rows, int rows_padded, uint16_t *buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over world * rows_padded
  if (i >= world * rows_padded) return;
  const int q = i / rows_padded, r = i - q * rows_padded;
  u32x4 w = {0u, 0u, 0u, 0u};
  if (q == rank && r < rows) {
    const float m = stats[2 * r];
    float mass = stats[2 * r + 1];
    if (!(mass > 0.f) || !(mass < INFINITY)) mass = 1.0f;
    int e;
    const float frac = frexpf(mass, &e);                       // mass = frac * 2^e, frac in [0.5, 1)
    const uint32_t mant = (uint32_t)(frac * 16777216.0f);       // 24 bits, exact
    const int64_t g = local_tokens[r] + (int64_t)rank * vocab_per_rank;
    auto f = [](float v) { return (uint32_t)elem<Tag>::st(v); };
    w[0] = f(m) | (f((float)(e + 64)) << 16);                  // (bias 64: masses below 1 do not occur, but stay exact)
    w[1] = f((float)(mant >> 16)) | (f((float)((mant >> 8) & 255u)) << 16);
    w[2] = f((float)(mant & 255u)) | (f((float)((g >> 16) & 255)) << 16);
    w[3] = f((float)((g >> 8) & 255)) | (f((float)(g & 255)) << 16);
  }
  *(u32x4 *)(buf + (int64_t)i * 8) = w;
}

template <typename Tag>
__global__ void mass_merge_kernel(const uint16_t *buf, const float *temperatures, uint64_t seed, uint64_t offset,
                                  const uint64_t *offset_dev, int world, int rows, int rows_padded, int64_t *tokens) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  auto ld = [](uint32_t half) { return elem<Tag>::ld((uint16_t)(half & 0xffffu)); };
  const float temp = temperatures ? temperatures[r] : 1.0f;
  float mq[16], sq[16];
  int64_t tq[16];
  float big = -INFINITY;
  for (int q = 0; q < world; ++q) {
    const u32x4 w = *(const u32x4 *)(buf + ((int64_t)q * rows_padded + r) * 8);
    mq[q] = ld(w[0]);
    const int e = (int)ld(w[0] >> 16) - 64;
    const uint32_t mant = ((uint32_t)ld(w[1]) << 16) | ((uint32_t)ld(w[1] >> 16) << 8) | (uint32_t)ld(w[2]);
    sq[q] = ldexpf((float)mant * (1.0f / 16777216.0f), e);
    tq[q] = ((int64_t)ld(w[2] >> 16) << 16) | ((int64_t)ld(w[3]) << 8) | (int64_t)ld(w[3] >> 16);
    if (mq[q] > big) big = mq[q];
  }
  int pick = 0;
  if (temp == 0.0f) {
    for (int q = 0; q < world; ++q)
      if (mq[q] == big) { pick = q; break; }                  // lowest rank among the maxima (NaN rows: rank 0)
  } else {
    const bool degenerate = !(big > -INFINITY) || !(big < INFINITY);
    const float inv_t = 1.4426950408889634f / fmaxf(temp, 1e-8f);
    float z = 0.f;
    for (int q = 0; q < world; ++q) {
      sq[q] = degenerate ? sq[q] : sq[q] * __builtin_amdgcn_exp2f((mq[q] - big) * inv_t);
      if (!(sq[q] >= 0.f)) sq[q] = 0.f;                        // a NaN maximum in one shard only
      z += sq[q];
    }
    const uint64_t off = offset + (offset_dev ? *offset_dev : 0ull);
    const float target = philox_uniform_b(seed, off, (uint32_t)r) * z;
    float acc = 0.f;
    pick = -1;
    int last = 0;
    for (int q = 0; q < world; ++q) {
      if (sq[q] > 0.f) {
        last = q;
        acc += sq[q];
        if (pick < 0 && target < acc) pick = q;
      }
    }
    if (pick < 0) pick = last;                                 // rounding at the very end
  }
  tokens[r] = tq[pick];
}


// ---- nucleus (top-p) / min-p rows of vocabulary-sharded logits -----------------------------------
// The edge of the nucleus is a property of the whole row: the smallest value v whose strictly-larger mass E(v) still
// fits p * Z.  The ranks find it together by a two-level radix walk over the 16-bit order keys, counted DOWN from the
// row's global maximum (offset = key(max) - key(x); level 0 bins the high byte of the offset, level 1 the low byte
// inside the bin level 0 chose), exchanging one 256-bin histogram of masses per level.  Masses are 2^-40 fixed point
// in 64-bit integers: integer sums do not depend on the order of the additions, so the LDS atomics, the all-reduce and
// therefore the edge are bit-identical on every rank, from run to run and for every TP degree.  (A weight below 2^-40
// of the maximum's counts as nothing: 150 k of them are 1e-7 of the row.)  The edge becomes a per-row floor; the
// ordinary sampler then runs on the shard with everything below the floor at -inf, and the two-stage draw above
// (mv_mass_pack / mv_mass_merge) picks the rank.
constexpr float kFix = 1099511627776.0f;  // 2^40

template <typename Tag>
__global__ void __launch_bounds__(kThreads) shard_key_hist_kernel(const uint16_t *logits, int64_t ld, int vocab_local,
                                                                  const float *row_max, const float *temperatures,
                                                                  const int32_t *prefix, unsigned long long *hist) {
  typedef typename elem<Tag>::T T;
  __shared__ unsigned long long h[256];
  const int rowi = blockIdx.x;
  if (threadIdx.x < 256) h[threadIdx.x] = 0ull;
  __syncthreads();
  const float m = row_max[rowi];
  const float temp = temperatures ? temperatures[rowi] : 1.0f;
  if (temp != 0.0f && m > -INFINITY && m < INFINITY) {  // block-uniform (greedy and degenerate rows: all zero)
    const T *row = (const T *)logits + (int64_t)rowi * ld;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
    const int kmax = (int)make_key<Tag>(elem<Tag>::st(m));
    const float inv_t = 1.4426950408889634f / fmaxf(temp, 1e-8f);
    const int want = prefix ? prefix[rowi] : -1;
    row_foreach<Tag>(row, vocab_local, vec_ok, [&](int i, T raw, bool ok) {
      int off = kmax - (int)make_key<Tag>(raw);
      off = off < 0 ? 0 : off;                         // -0.0 against a +0.0 maximum
      const float w = __builtin_amdgcn_exp2f((elem<Tag>::ld(raw) - m) * inv_t);
      const bool in = ok && (w == w) && (want < 0 || (off >> 8) == want);
      const unsigned bin = want < 0 ? (unsigned)(off >> 8) : (unsigned)(off & 255);
      if (in) atomicAdd(&h[bin], (unsigned long long)(w * kFix));
    });
  }
  __syncthreads();
  if (threadIdx.x < 256) hist[(int64_t)rowi * 256 + threadIdx.x] = h[threadIdx.x];
}

// One 256-thread block per row walks a summed histogram.  Level 0: budget = p * Z, prefix = the last high-byte bin
// whose strictly-larger mass fits it, state = {budget, that mass}.  Level 1: the same walk over the low byte, starting
// from the carried mass; the value that owns the chosen key is the nucleus floor, raised to the min-p floor
// (weight >= min_p  <=>  x >= max + T ln(min_p)) when that one is higher.  Rows without either filter get -inf.
template <typename Tag>
__global__ void __launch_bounds__(256) nucleus_pick_kernel(const unsigned long long *hist, int level, const float *top_ps,
                                                           const float *min_ps, const float *temperatures,
                                                           const float *row_max, unsigned long long *state,
                                                           int32_t *prefix, float *floor_out) {
  typedef typename elem<Tag>::T T;
  __shared__ unsigned long long scan[256];
  __shared__ unsigned long long carry[2];
  const int rowi = blockIdx.x, t = threadIdx.x;
  const unsigned long long own = hist[(int64_t)rowi * 256 + t];
  scan[t] = own;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {                      // inclusive Hillis-Steele scan
    const unsigned long long add = t >= d ? scan[t - d] : 0ull;
    __syncthreads();
    scan[t] += add;
    __syncthreads();
  }
  const unsigned long long before = scan[t] - own;         // mass of the bins above mine
  const float top_p = top_ps ? top_ps[rowi] : 1.0f;
  unsigned long long budget, base;
  if (level == 0) {
    const double z = (double)scan[255];
    budget = (unsigned long long)(fmin(fmax((double)top_p, 0.0), 1.0) * z);
    base = 0ull;
  } else {
    budget = state[2 * rowi];
    base = state[2 * rowi + 1];
  }
  const int fits = __syncthreads_count(base + before <= budget);   // a prefix of the bins: bin 0 always fits
  const int pick = fits > 0 ? fits - 1 : 0;
  if (level == 0) {
    if (t == pick) {
      state[2 * rowi] = budget;
      state[2 * rowi + 1] = before;
      prefix[rowi] = pick;
    }
    return;
  }
  if (t != pick) return;
  const float temp = temperatures ? temperatures[rowi] : 1.0f;
  const float m = row_max[rowi];
  float fl = -INFINITY;
  if (temp != 0.0f && m > -INFINITY && m < INFINITY) {
    if (top_p < 1.0f) {
      int key = (int)make_key<Tag>(elem<Tag>::st(m)) - (prefix[rowi] * 256 + pick);
      key = key < 0 ? 0 : key;
      const uint32_t rawbits = (key & 0x8000) ? (uint32_t)(key & 0x7fff) : (~(uint32_t)key & 0xffffu);
      fl = elem<Tag>::ld((T)rawbits);
      if (!(fl == fl)) fl = -INFINITY;                       // the key ran into the NaN encodings: keep everything
    }
    const float min_p = min_ps ? min_ps[rowi] : 0.0f;
    if (min_p > 0.0f) fl = fmaxf(fl, m + fmaxf(temp, 1e-8f) * __logf(fminf(min_p, 1.0f)));
  }
  floor_out[rowi] = fl;
}

template <typename Tag>
__global__ void __launch_bounds__(256) mask_below_kernel(const uint16_t *in, int64_t ld_in, const float *floor_row,
                                                         uint16_t *out, int64_t ld_out, int vocab_local, bool vec_ok) {
  typedef typename elem<Tag>::T T;
  const int rowi = blockIdx.y;
  const float fl = floor_row[rowi];
  const T ninf = elem<Tag>::st(-INFINITY);
  const uint16_t *src = in + (int64_t)rowi * ld_in;
  uint16_t *dst = out + (int64_t)rowi * ld_out;
  auto keep = [&](uint32_t half) -> uint32_t {
    const float x = elem<Tag>::ld((T)(half & 0xffffu));
    return (x >= fl || !(x == x)) ? (half & 0xffffu) : (uint32_t)ninf;   // NaN stays NaN: the sampler's business
  };
  if (vec_ok) {
    const int nvec = vocab_local / 8;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < nvec; c += gridDim.x * 256) {
      u32x4 v = *(const u32x4 *)(src + (int64_t)c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = keep(v[j]) | (keep(v[j] >> 16) << 16);
      *(u32x4 *)(dst + (int64_t)c * 8) = v;
    }
    for (int i = nvec * 8 + blockIdx.x * 256 + threadIdx.x; i < vocab_local; i += gridDim.x * 256)
      dst[i] = (uint16_t)keep(src[i]);
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < vocab_local; i += gridDim.x * 256) dst[i] = (uint16_t)keep(src[i]);
  }
}

}  // namespace

extern "C" int mv_mass_pack(const float *stats, const int64_t *local_tokens, int rank, int world,
                            int64_t vocab_per_rank, int rows, int rows_padded, void *buf, int dtype, void *stream) {
  if (rows == 0) return MV_OK;
  if (!stats || !local_tokens || !buf) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (rows < 0 || rows_padded < rows || world < 1 || world > 16 || rank < 0 || rank >= world ||
      vocab_per_rank * world > (1 << 24))
    return MV_E_SHAPE;
  if (!mv_aligned16(buf)) return MV_E_ALIGN;
  const int n = world * rows_padded;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((mass_pack_kernel<bf16_tag>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats,
                       local_tokens, rank, world, vocab_per_rank, rows, rows_padded, (uint16_t *)buf);
  else
    hipLaunchKernelGGL((mass_pack_kernel<f16_tag>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats,
                       local_tokens, rank, world, vocab_per_rank, rows, rows_padded, (uint16_t *)buf);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_mass_merge(const void *buf, const float *temperatures, uint64_t seed, uint64_t offset,
                             const uint64_t *offset_dev, int world, int rows, int rows_padded, int64_t *tokens,
                             int dtype, void *stream) {
  if (rows == 0) return MV_OK;
  if (!buf || !tokens) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (rows < 0 || rows_padded < rows || world < 1 || world > 16) return MV_E_SHAPE;
  if (!mv_aligned16(buf)) return MV_E_ALIGN;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((mass_merge_kernel<bf16_tag>), dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)buf, temperatures, seed, offset, offset_dev, world, rows, rows_padded, tokens);
  else
    hipLaunchKernelGGL((mass_merge_kernel<f16_tag>), dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)buf, temperatures, seed, offset, offset_dev, world, rows, rows_padded, tokens);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_shard_key_hist(const void *logits_local, int64_t ld, int vocab_local, int rows, const float *row_max,
                                 const float *temperatures, const int32_t *prefix, int64_t *hist, int dtype,
                                 void *stream) {
  if (rows == 0) return MV_OK;
  if (!logits_local || !row_max || !hist) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (rows < 0 || vocab_local <= 0 || ld < vocab_local) return MV_E_SHAPE;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((shard_key_hist_kernel<bf16_tag>), dim3(rows), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t *)logits_local, ld, vocab_local, row_max, temperatures, prefix,
                       (unsigned long long *)hist);
  else
    hipLaunchKernelGGL((shard_key_hist_kernel<f16_tag>), dim3(rows), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t *)logits_local, ld, vocab_local, row_max, temperatures, prefix,
                       (unsigned long long *)hist);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_nucleus_pick(const int64_t *hist, int level, const float *top_ps, const float *min_ps,
                               const float *temperatures, const float *row_max, int64_t *state, int32_t *prefix,
                               float *floor_out, int rows, int dtype, void *stream) {
  if (rows == 0) return MV_OK;
  if (!hist || !state || !prefix || (level == 1 && (!floor_out || !row_max))) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (rows < 0 || (level != 0 && level != 1)) return MV_E_SHAPE;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((nucleus_pick_kernel<bf16_tag>), dim3(rows), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long *)hist, level, top_ps, min_ps, temperatures, row_max,
                       (unsigned long long *)state, prefix, floor_out);
  else
    hipLaunchKernelGGL((nucleus_pick_kernel<f16_tag>), dim3(rows), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long *)hist, level, top_ps, min_ps, temperatures, row_max,
                       (unsigned long long *)state, prefix, floor_out);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_mask_below(const void *logits, int64_t ld, int vocab_local, int rows, const float *floor_row,
                             void *out, int64_t ld_out, int dtype, void *stream) {
  if (rows == 0) return MV_OK;
  if (!logits || !floor_row || !out) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (rows < 0 || rows > 65535 || vocab_local <= 0 || ld < vocab_local || ld_out < vocab_local) return MV_E_SHAPE;
  const bool vec_ok = mv_aligned16(logits) && mv_aligned16(out) && ld % 8 == 0 && ld_out % 8 == 0;
  const int gx = (vocab_local / 8 + 255) / 256 > 0 ? (vocab_local / 8 + 255) / 256 : 1;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((mask_below_kernel<bf16_tag>), dim3(gx, rows), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)logits, ld, floor_row, (uint16_t *)out, ld_out, vocab_local, vec_ok);
  else
    hipLaunchKernelGGL((mask_below_kernel<f16_tag>), dim3(gx, rows), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)logits, ld, floor_row, (uint16_t *)out, ld_out, vocab_local, vec_ok);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_topc_pick(const void *cand_logits, const int32_t *cand_idx, const int64_t *pos,
                            const float *temperatures, int64_t *tokens, int rows, int width, int dtype, void *stream) {
  if (rows == 0) return MV_OK;
  if (!cand_logits || !cand_idx || !pos || !tokens) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (rows < 0 || width <= 0) return MV_E_SHAPE;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((topc_pick_kernel<bf16_tag>), dim3(rows), dim3(64), 0, (hipStream_t)stream,
                       (const uint16_t *)cand_logits, cand_idx, pos, temperatures, tokens, width);
  else
    hipLaunchKernelGGL((topc_pick_kernel<f16_tag>), dim3(rows), dim3(64), 0, (hipStream_t)stream,
                       (const uint16_t *)cand_logits, cand_idx, pos, temperatures, tokens, width);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" size_t mv_sample_workspace_bytes(int batch, int vocab) {
  (void)batch; (void)vocab;
  return 256;  // the kernel keeps everything in LDS/registers; reserved for future use
}

extern "C" int mv_sample(const void *logits, int64_t ld, const float *temperatures,
                         const int64_t *top_ks, const float *top_ps, const float *min_ps,
                         uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
                         int64_t *out_tokens, uint8_t *keep_mask, int batch, int vocab,
                         void *workspace, size_t ws_bytes, int dtype, void *stream) {
  if (batch == 0) return MV_OK;
  if (!logits || !out_tokens) return MV_E_NULLPTR;
  if (batch < 0 || vocab <= 0 || ld < vocab) return MV_E_SHAPE;
  if (workspace && (((uintptr_t)workspace) & 3u)) return MV_E_ALIGN;
  SampleParams p;
  // a workspace of >= 8 bytes per row receives {row maximum, kept mass} (fp32 pairs): what the tensor-parallel
  // two-stage draw (mv_mass_pack / mv_mass_merge) exchanges instead of the logits
  p.stats = (workspace && ws_bytes >= (size_t)batch * 8) ? (float *)workspace : nullptr;
  p.logits = logits; p.ld = ld; p.temperatures = temperatures; p.top_ks = top_ks;
  p.top_ps = top_ps; p.min_ps = min_ps; p.seed = seed; p.offset = offset;
  p.offset_dev = offset_dev; p.out_tokens = out_tokens; p.keep_mask = keep_mask; p.vocab = vocab;
  MV_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((sample_kernel<Tag>), dim3(batch), dim3(kThreads), 0,
                                              (hipStream_t)stream, p));
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_topc_width(void) { return kTopC; }

extern "C" int mv_topc_pack(const void *logits_local, int64_t ld, int vocab_local, int rank, int world,
                            int64_t vocab_per_rank, int rows, int rows_padded, void *buf, int dtype, void *stream) {
  if (rows == 0) return MV_OK;
  if (!logits_local || !buf) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (rows < 0 || rows_padded < rows || world < 1 || rank < 0 || rank >= world || vocab_local <= 0 || ld < vocab_local ||
      vocab_per_rank * world > (1 << 24))
    return MV_E_SHAPE;
  if (((uintptr_t)buf) & 7u) return MV_E_ALIGN;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((topc_pack_kernel<bf16_tag>), dim3(rows), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t *)logits_local, ld, vocab_local, rank, world, vocab_per_rank, rows_padded, (uint16_t *)buf);
  else
    hipLaunchKernelGGL((topc_pack_kernel<f16_tag>), dim3(rows), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t *)logits_local, ld, vocab_local, rank, world, vocab_per_rank, rows_padded, (uint16_t *)buf);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_topc_unpack(const void *buf, int world, int rows, int rows_padded, void *cand_logits, int32_t *cand_idx,
                              int dtype, void *stream) {
  if (rows == 0) return MV_OK;
  if (!buf || !cand_logits || !cand_idx) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (rows < 0 || rows_padded < rows || world < 1) return MV_E_SHAPE;
  const int n = rows * world * kTopC;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((topc_unpack_kernel<bf16_tag>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)buf, world, rows, rows_padded, (uint16_t *)cand_logits, cand_idx);
  else
    hipLaunchKernelGGL((topc_unpack_kernel<f16_tag>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)buf, world, rows, rows_padded, (uint16_t *)cand_logits, cand_idx);
  MV_LAUNCH_CHECK();
  return MV_OK;
}
