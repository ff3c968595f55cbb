// K12 — the decoder layers of a small decode step (<= 16 rows) as ONE persistent launch.
//
// Op order (reference: minivllm/models/qwen_base.py:173-190 attention block, :232-236 MLP, :281-298 layer):
//   A  residual add + RMSNorm (input_layernorm) -> qkv projection (+ bias)
//   B  q/k head RMSNorm -> RoPE -> K/V store -> paged attention (context split over workgroups)
//   C  split-KV combine
//   D  o_proj
//   E  residual add + RMSNorm (post_attention_layernorm) -> gate_up projection -> SiluAndMul
//   F  down_proj
// for every layer, between grid barriers — what the multi-launch path runs as six launches per layer
// (mv_norm_linear_small_m_packed, mv_rope_paged_decode + combine, mv_linear_small_m_packed,
// mv_norm_gate_up_silu_small_m_packed, mv_linear_small_m_packed).  Why one launch: at these batch sizes a layer is ~100 MB of
// weights (16 us of HBM time) behind six launches of ~3.5 us fixed cost each; here the fixed cost of an edge is the grid
// barrier (2.2 us fence-free, profiles/r04_barrier_probe.md) and the next projection's weights are ALREADY IN FLIGHT while the
// workgroup waits at it: waves 1..7 issue their first weight chunks by LDS-DMA right after their own arrival, wave 0 (which
// polls) issues none, so that its poll never queues behind a DMA (vector-memory returns are in order per wave).
//
// Arithmetic: the bodies are the multi-launch kernels' own (paged_decode_body.h) or restate them operation for operation
// (norm prologue = rmsnorm_kernel thread for thread; projections = linear_small_m_kernel<PACKED>: the same K slices — one per
// "virtual wave" of that kernel's launch plan —, the same v_mfma_f32_16x16x32 chain per slice, partial tiles summed in slice
// order, one rounding).  Outputs are bit-identical to the six-launch path (tests/test_gpu_persistent.py).
//
// Cross-workgroup visibility is the fence-free form of MI355X_MICROARCH.md ("Valid forms", table row 1): every activation
// byte that another workgroup reads is stored write-through and loaded with an sc1 load (paged_decode.h: ld16<true> ...),
// the storing waves drain and meet at a workgroup barrier before lane 0 arrives.  Every spin is bounded; a give-up is
// recorded in the status word and every later barrier returns at once (the launch drains, its results are garbage, the host
// raises).  The barrier state is zeroed by a one-workgroup launch in front.  One workgroup per CU, all resident.
#include "paged_decode_body.h"

#include <mutex>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

template <typename Tag>
__device__ __forceinline__ f32x4 mfma16(const u32x4 &a, const u32x4 &b, const f32x4 &c) {
  if constexpr (__is_same(Tag, bf16_tag))
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// pointers that come out of memory (the layer table) or through structs lose their address space: say it again, or every
// access through them becomes a flat_ instruction (out-of-order returns, both counters)
template <typename T>
__device__ __forceinline__ T *gptr(T *p) { return (T *)(__attribute__((address_space(1))) T *)p; }
template <typename T>
__device__ __forceinline__ T *lptr(T *p) { return (T *)(__attribute__((address_space(3))) T *)p; }

constexpr int kThreads = 512, kWaves = 8, kStreamWaves = 7;   // wave 0 polls the grid barrier and streams nothing
constexpr int kChunkSteps = 8;                                  // k-steps (32 elements) of one weight tile per chunk
constexpr int kChunkBytes = kChunkSteps * 1024;                 // 8 KiB: 16 rows x 256 elements
constexpr int kMaxSlices = 16;

struct BarState {             // one 128-byte line per word that is polled or added to; zeroed before every launch
  unsigned census[8][32];     // workgroups per XCC
  unsigned flat[32];          // census barrier
  unsigned xcd[8][32];        // per-XCC arrival counters
  unsigned top[32];
  unsigned gen[8][32];        // per-XCC generation
  unsigned giveup[32];        // != 0: a bounded wait expired somewhere; every wait returns at once from then on
};

struct Proj {                 // one projection kind (the same for every layer; the weights come from the layer table)
  int n, n_tiles, K;          // output columns (gated: of the intermediate size), ceil(n / 16), input features
  int nslices, kU;            // K slices ("virtual waves" of linear_small_m_kernel's launch plan) and its block size in k-steps
  int cs;                     // k-steps per streamed chunk: 8, or 4 where a slice is not a multiple of 8 steps
};

struct Params {
  const MvPersistLayer *layers;
  int n_layers, M;
  int hidden, inter, nqkv, q_size, kv_size;
  float eps;
  Proj qkv, o, gu, down;
  int norm_threads, norm_pt;           // rmsnorm_kernel's block size / vectors per thread for `hidden` columns
  int npre;                            // weight chunks per streaming wave in flight across a barrier (0 = no prefetch)
  const void *x_in, *res_in;           // layer 0: hidden states (and residual, or null)
  int64_t x_in_stride, res_in_stride;
  char *ws;                            // workspace: activations between phases (offsets below), barrier state in front
  int64_t off_res0, off_res1, off_h, off_qkv, off_attn, off_o, off_act, off_wsacc, off_wsml;
  DecodeParams dp;                     // attention: everything but the per-layer pointers
  int att_gx, att_gy;                  // grid of the decode kernel (kv-head groups, splits); z = M
  unsigned *status;                    // [0] give-ups, [1..3] record of the first wait that expired (spin_until_ge)
  int stamp_layer;                     // >= 0: workgroup 1 stamps s_memrealtime at the phase edges of this layer into status[8..]
};

__device__ __forceinline__ unsigned ld_rlx(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rlx(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_rlx(unsigned *p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Bounded in WALL-CLOCK time (s_memrealtime, 100 MHz): 2 s.  The first wait that expires records who / where / what it saw
// (status words 1..3: epoch, workgroup, value seen) and raises the give-up word; every later wait returns at once.
__device__ __forceinline__ void spin_until_ge(unsigned *p, unsigned want, BarState *st, unsigned *status, unsigned where) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  unsigned seen = 0;
  for (unsigned it = 0;; ++it) {
    if ((seen = ld_rlx(p)) >= want) return;
    if ((it & 63) == 63) {
      if (ld_rlx(st->giveup) != 0) return;
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  if (__hip_atomic_exchange(st->giveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
    status[1] = where, status[2] = want, status[3] = seen;
  }
}

// LDS-DMA of one 1-KiB weight fragment (64 lanes x 16 B, contiguous in the packed copy) into this wave's ring
__device__ __forceinline__ void dma_1k(const void *gsrc_lane, char *lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                   (__attribute__((address_space(3))) void *)lds_dst_wave_uniform, 16, 0, /*nt*/ 2);
}

struct Ctx {                   // per-workgroup constants
  BarState *bar;
  unsigned xcc, xcd_size, n_xcd, epoch;
  int nwg, wg, tid, wave, lane;
  unsigned *status;
  char *ring;                  // [kStreamWaves][npre][kChunkBytes]
  char *scratch;               // phase-local LDS behind the ring
};

// K slice s of a projection, in k-steps: linear_small_m_kernel<PACKED>'s split (whole blocks of kU steps per slice)
__device__ __forceinline__ void slice_range(const Proj &pj, int s, int &lo, int &hi) {
  const int steps = pj.K >> 5;
  const int nblk = (steps + pj.kU - 1) / pj.kU;
  lo = (int)((int64_t)nblk * s / pj.nslices) * pj.kU;
  hi = min(steps, (int)((int64_t)nblk * (s + 1) / pj.nslices) * pj.kU);
}

// The weight chunks of a streaming wave in one projection phase, in the order it consumes them:
//   for every output tile v of this workgroup, every slice s of this wave, every block of <= 8 k-steps, every tile a of the
//   pair (GATED: gate tile, up tile).  `fn(j, tile_index_in_packed_copy, k0, nsteps)`; returns the chunk count.
template <bool GATED, typename F>
__device__ __forceinline__ int for_each_chunk(const Proj &pj, const Ctx &c, int limit, F fn) {
  int j = 0;
  if (c.wave == 0) return 0;
  for (int v = c.wg; v < pj.n_tiles; v += c.nwg) {
    for (int s = c.wave - 1; s < pj.nslices; s += kStreamWaves) {
      int lo, hi;
      slice_range(pj, s, lo, hi);
      for (int k0 = lo; k0 < hi; k0 += pj.cs) {
#pragma unroll
        for (int a = 0; a < (GATED ? 2 : 1); ++a) {
          if (j >= limit) return j;
          fn(j, v + (a ? pj.n_tiles : 0), k0, pj.cs);
          ++j;
        }
      }
    }
  }
  return j;
}

template <bool GATED>
__device__ __forceinline__ void issue_prefetch(const Proj &pj, const void *w_packed, const Ctx &c, int npre) {
  if (npre <= 0) return;
  const int steps = pj.K >> 5;
  MV_AS3 char *ring = (MV_AS3 char *)c.ring + (size_t)(c.wave - 1) * npre * kChunkBytes;
  for_each_chunk<GATED>(pj, c, npre, [&](int j, int tile, int k0, int ns) {
    const MV_AS1 uint16_t *src = (const MV_AS1 uint16_t *)w_packed + ((int64_t)tile * steps + k0) * 512 + c.lane * 8;
    for (int u = 0; u < ns; ++u)
      __builtin_amdgcn_global_load_lds((const MV_AS1 void *)(src + (int64_t)u * 512), (MV_AS3 void *)(ring + j * kChunkBytes + u * 1024),
                                       16, 0, /*nt*/ 2);
  });
}

// ---- grid barrier (XCD-hierarchical, fence-free: see the header) + the next projection's first weight chunks --------------
template <typename PF>
__device__ __forceinline__ void grid_barrier(Ctx &c, PF prefetch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its write-through stores have left
  __syncthreads();
  ++c.epoch;
  if (c.wave != 0) {
    prefetch();                                       // LDS-DMA, not waited for here
  } else if (c.lane == 0) {
    BarState *st = c.bar;
    const unsigned n = add_rlx(st->xcd[c.xcc], 1u) + 1;
    if (n == c.xcd_size * c.epoch) {
      const unsigned t = add_rlx(st->top, 1u) + 1;
      if (t == c.n_xcd * c.epoch) {
        for (unsigned x = 0; x < 8; ++x) st_rlx(st->gen[x], c.epoch);
      }
    }
    spin_until_ge(st->gen[c.xcc], c.epoch, st, c.status, (c.epoch << 16) | (unsigned)c.wg);
  }
  __syncthreads();
}

constexpr int norm_row_bytes(int k) { return k * 2 + 64; }

// ---- residual add + RMSNorm of the <= 16 rows, by every workgroup (rmsnorm_kernel's arithmetic thread for thread: a block
// of T = norm_threads threads per row, thread t owning the 16-byte vectors t, t + T, ...; fmaf chain, wave_sum, wave-order
// block sum; the norm of the UNROUNDED sum, the rounded sum = new residual, written by workgroup 0) -> normalised rows in LDS.
// Two rows at a time: threads 0..255 and 256..511 each play one block of T <= 256 threads.
template <typename Tag, int PT>
__device__ __forceinline__ void norm_rows(const Params &p, const Ctx &c, const void *x, int64_t x_stride, const void *res,
                                          int64_t res_stride, void *res_out, const void *normw, float *nred_, char *xs_,
                                          bool x_coherent) {
  const int T = p.norm_threads, K = p.hidden, nvec = K >> 3, nw = T >> 6;
  const int half = c.tid >> 8, t = c.tid & 255;
  const uint16_t *xp = (const uint16_t *)x, *rp = (const uint16_t *)res;
  const MV_AS1 uint16_t *w = (const MV_AS1 uint16_t *)normw;
  uint16_t *ro = (uint16_t *)res_out;
  MV_AS3 float *nred = (MV_AS3 float *)nred_;
  MV_AS3 char *xs = (MV_AS3 char *)xs_;
  const bool add = res != nullptr;
  u32x4 wraw[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int col = t + i * T;
    wraw[i] = (t < T && col < nvec) ? *(const MV_AS1 u32x4 *)(w + col * 8) : u32x4{0u, 0u, 0u, 0u};
  }
  for (int r0 = 0; r0 < p.M; r0 += 2) {
    const int r = r0 + half;
    const bool row_on = r < p.M && t < T;
    float vals[PT][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int col = t + i * T;
      const bool on = row_on && col < nvec;
      u32x4 xr = {0u, 0u, 0u, 0u}, rr = {0u, 0u, 0u, 0u};
      if (on) {
        xr = x_coherent ? ld16<true>(xp + (int64_t)r * x_stride + col * 8) : ld16<false>(xp + (int64_t)r * x_stride + col * 8);
        if (add) rr = x_coherent ? ld16<true>(rp + (int64_t)r * res_stride + col * 8) : ld16<false>(rp + (int64_t)r * res_stride + col * 8);
      }
      unpack8<Tag>(xr, vals[i]);
      if (add) {
        float rv[8];
        unpack8<Tag>(rr, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) vals[i][j] = __fadd_rn(vals[i][j], rv[j]);
      }
      if (on && c.wg == 0) st16<true>(ro + (int64_t)r * K + col * 8, add ? pack8<Tag>(vals[i]) : xr);
      if (on) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ss = fmaf(vals[i][j], vals[i][j], ss);
      }
    }
    ss = wave_sum(ss);
    if ((t & 63) == 0 && (t >> 6) < nw && r < p.M) nred[r * 16 + (t >> 6)] = ss;
    __syncthreads();
    if (row_on) {
      float tot = nred[r * 16];
      for (int i = 1; i < nw; ++i) tot += nred[r * 16 + i];
      const float var = tot / (float)K;
      const float rs = 1.0f / sqrtf(var + p.eps);
#pragma unroll
      for (int i = 0; i < PT; ++i) {
        const int col = t + i * T;
        if (col < nvec) {
          float wv[8], y[8];
          unpack8<Tag>(wraw[i], wv);
#pragma unroll
          for (int j = 0; j < 8; ++j) y[j] = __fmul_rn(__fmul_rn(vals[i][j], rs), wv[j]);
          *(MV_AS3 u32x4 *)(xs + r * norm_row_bytes(K) + col * 16) = pack8<Tag>(y);
        }
      }
    }
  }
  __syncthreads();
}

// ---- one projection phase: y[M, n] = x[M, K] . W^T (W = the packed copy), for the output tiles of this workgroup ---------
// NORM: x = the normalised rows in LDS (xs).  else: x rows in the workspace (coherent loads).  GATED: tiles (v, v + n_tiles)
// = gate / up rows of the same output columns, SiluAndMul in the epilogue.
template <typename Tag, bool NORM, bool GATED, int CS>
__device__ __forceinline__ void projection_cs(const Params &p, const Ctx &c, const Proj &pj, const void *w_packed_, const void *bias_,
                                              const char *xs_, const void *x_, int64_t x_stride, void *out_, int64_t out_stride,
                                              f32x4 *red4_, int npre) {
  typedef uint16_t T;
  constexpr int NA = GATED ? 2 : 1;
  const MV_AS1 T *w_packed = (const MV_AS1 T *)w_packed_, *bias = (const MV_AS1 T *)bias_;
  const T *x = (const T *)x_;
  T *out = (T *)out_;
  const MV_AS3 char *xs = (const MV_AS3 char *)xs_;
  MV_AS3 f32x4 *red4 = (MV_AS3 f32x4 *)red4_;
  const int lane = c.lane, r16 = lane & 15, g = lane >> 4;
  const int steps = pj.K >> 5;
  const MV_AS3 char *ring = (const MV_AS3 char *)c.ring + (size_t)(c.wave > 0 ? c.wave - 1 : 0) * npre * kChunkBytes;
  const int mrow = r16 < p.M ? r16 : p.M - 1;
  const MV_AS3 char *xs_row = xs + mrow * norm_row_bytes(pj.K) + g * 16;
  const T *xg = NORM ? nullptr : x + (int64_t)mrow * x_stride + g * 8;
  bool first = true;
  int j = 0;
  for (int v = c.wg; v < pj.n_tiles; v += c.nwg) {
    if (c.wave != 0) {
      for (int s = c.wave - 1; s < pj.nslices; s += kStreamWaves) {
        int lo, hi;
        slice_range(pj, s, lo, hi);
        f32x4 acc[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k0 = lo; k0 < hi; k0 += CS) {
          u32x4 wf[NA][CS], xf[CS];
          // x first: its loads must not queue behind weight loads issued now (in-order returns); the prefetched chunks are
          // older than both
#pragma unroll
          for (int u = 0; u < CS; ++u) {
            if constexpr (NORM) xf[u] = *(const MV_AS3 u32x4 *)(xs_row + (k0 + u) * 64);
            else xf[u] = ld16<true>(xg + (k0 + u) * 32);
          }
          if (j < npre && first) {
            // the chunks the barrier in front of this phase left in flight: the DMA is invisible to the compiler's counting
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            first = false;
          }
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            const int tile = v + (a ? pj.n_tiles : 0);
            if (j < npre) {
#pragma unroll
              for (int u = 0; u < CS; ++u) wf[a][u] = *(const MV_AS3 u32x4 *)(ring + j * kChunkBytes + u * 1024 + lane * 16);
            } else {
              const MV_AS1 T *src = w_packed + ((int64_t)tile * steps + k0) * 512 + lane * 8;
#pragma unroll
              for (int u = 0; u < CS; ++u) wf[a][u] = __builtin_nontemporal_load((const MV_AS1 u32x4 *)(src + (int64_t)u * 512));
            }
            ++j;
          }
#pragma unroll
          for (int u = 0; u < CS; ++u) {
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[a] = mfma16<Tag>(wf[a][u], xf[u], acc[a]);
          }
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) red4[(s * NA + a) * 64 + lane] = acc[a];
      }
    }
    __syncthreads();
    // finished by wave 1: partial tiles summed in slice order (linear_small_m_kernel's tile_sum), one rounding
    if (c.wave == 1) {
      auto tile_sum = [&](int a) {
        f32x4 sum = red4[a * 64 + lane];
        for (int s = 1; s < pj.nslices; ++s) {
          const f32x4 t = red4[(s * NA + a) * 64 + lane];
          sum[0] += t[0]; sum[1] += t[1]; sum[2] += t[2]; sum[3] += t[3];
        }
        return sum;
      };
      const int m = r16, n = v * 16 + g * 4;
      const f32x4 sum = tile_sum(0);
      float o[4] = {sum[0], sum[1], sum[2], sum[3]};
      if constexpr (GATED) {
        const f32x4 up = tile_sum(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gt = elem<Tag>::ld(elem<Tag>::st(o[i]));
          const float ut = elem<Tag>::ld(elem<Tag>::st(up[i]));
          float sl = gt / (1.0f + expf(-gt));
          sl = elem<Tag>::ld(elem<Tag>::st(sl));
          o[i] = __fmul_rn(sl, ut);
        }
      }
      if (m < p.M && n < pj.n) {
        if (!GATED && bias) {
          const MV_AS1 T *bp = bias + n;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] += elem<Tag>::ld(bp[i]);
        }
        u32x2 pk;
        pk[0] = (uint32_t)elem<Tag>::st(o[0]) | ((uint32_t)elem<Tag>::st(o[1]) << 16);
        pk[1] = (uint32_t)elem<Tag>::st(o[2]) | ((uint32_t)elem<Tag>::st(o[3]) << 16);
        st8<true>(out + (int64_t)m * out_stride + n, pk);
      }
    }
    __syncthreads();
  }
}

template <typename Tag, bool NORM, bool GATED>
__device__ __forceinline__ void projection(const Params &p, const Ctx &c, const Proj &pj, const void *w_packed, const void *bias,
                                           const char *xs, const void *x, int64_t x_stride, void *out, int64_t out_stride,
                                           f32x4 *red4, int npre) {
  if (pj.cs == 8) projection_cs<Tag, NORM, GATED, 8>(p, c, pj, w_packed, bias, xs, x, x_stride, out, out_stride, red4, npre);
  else projection_cs<Tag, NORM, GATED, 4>(p, c, pj, w_packed, bias, xs, x, x_stride, out, out_stride, red4, npre);
}

// The barrier state is zeroed by a launch of its own in front of the persistent kernel.  (A hipMemsetAsync node did this in the
// first version: replayed from a captured graph on ROCm 7.2 it filled the block with the low word of an address instead
// of 0 — every counter started at 0xd0252800 — found with tools/dbg/persist_dbg.py; eager launches were fine.)
__global__ void __launch_bounds__(256) zero_state_kernel(unsigned *words, int n) {
  for (int i = threadIdx.x; i < n; i += 256) words[i] = 0u;
}

template <typename Tag, int D, int G, bool WIDE>
__global__ void __launch_bounds__(kThreads) decode_layers_persistent_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ __attribute__((aligned(16))) unsigned s_info[4];   // (16 bytes: the dynamic region behind it stays 16-byte aligned)
  unsigned &s_xcc = s_info[0], &s_size = s_info[1], &s_nx = s_info[2];
  Ctx c;
  c.bar = (BarState *)p.ws;
  c.nwg = gridDim.x, c.wg = blockIdx.x, c.tid = threadIdx.x, c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.lane = threadIdx.x & 63;
  c.epoch = 0;
  c.status = p.status;
  c.ring = smem;
  c.scratch = smem + (size_t)kStreamWaves * p.npre * kChunkBytes;
  // census: workgroups per XCC (placement is not promised by HIP: counted, not assumed)
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7;   // HW_REG_XCC_ID
    s_xcc = xcc;
    add_rlx(c.bar->census[xcc], 1u);
    add_rlx(c.bar->flat, 1u);
    spin_until_ge(c.bar->flat, (unsigned)c.nwg, c.bar, p.status, 0xffff0000u | (unsigned)blockIdx.x);
    unsigned nx = 0;
    for (unsigned x = 0; x < 8; ++x) nx += ld_rlx(c.bar->census[x]) != 0;
    s_size = ld_rlx(c.bar->census[xcc]);
    s_nx = nx;
  }
  // the first layer's qkv weights do not wait for the census
  if (c.wave != 0) issue_prefetch<false>(p.qkv, gptr(p.layers)[0].wqkv, c, p.npre);
  __syncthreads();
  c.xcc = s_xcc, c.xcd_size = s_size, c.n_xcd = s_nx;

  float *nred = (float *)c.scratch;                       // [16 rows][16 waves]
  char *xs = c.scratch + 1024;                             // normalised rows (NORM phases)
  f32x4 *red4 = (f32x4 *)(c.scratch + 1024 + (size_t)p.M * norm_row_bytes(p.hidden));   // partial tiles [slices][NA][64]
  float *att_lds = lptr((float *)(c.scratch + 1024));      // attention / combine slots (phases B, C)
  char *ws = gptr(p.ws);
  const int K = p.hidden;
  auto none = [] {};

  // diagnostic (flags bit 8): wall-clock stamps (100 MHz) of one workgroup at the phase edges of one layer, low 32 bits, into
  // status[8 + i] — memory nothing else reads
  auto stamp = [&](int l, int i) {
    if (p.stamp_layer == l && c.wg == 1 && threadIdx.x == 64) p.status[8 + i] = (unsigned)__builtin_amdgcn_s_memrealtime();
  };
  for (int l = 0; l < p.n_layers; ++l) {
    MvPersistLayer L = gptr(p.layers)[l];
    L.ln1_w = gptr(L.ln1_w), L.wqkv = gptr(L.wqkv), L.bqkv = gptr(L.bqkv), L.q_norm_w = gptr(L.q_norm_w);
    L.k_norm_w = gptr(L.k_norm_w), L.wo = gptr(L.wo), L.ln2_w = gptr(L.ln2_w), L.wgu = gptr(L.wgu), L.wdown = gptr(L.wdown);
    L.k_cache = gptr(L.k_cache), L.v_cache = gptr(L.v_cache);
    // ---- A: input norm + qkv
    stamp(l, 0);
    if (l > 0) grid_barrier(c, [&] { issue_prefetch<false>(p.qkv, L.wqkv, c, p.npre); });
    stamp(l, 1);
    {
      const void *x = l == 0 ? p.x_in : ws + p.off_h;
      const int64_t xst = l == 0 ? p.x_in_stride : K;
      const void *res = l == 0 ? p.res_in : ws + p.off_res1;
      const int64_t rst = l == 0 ? p.res_in_stride : K;
      if (p.norm_pt <= 1) norm_rows<Tag, 1>(p, c, x, xst, res, rst, ws + p.off_res0, L.ln1_w, nred, xs, l > 0);
      else if (p.norm_pt == 2) norm_rows<Tag, 2>(p, c, x, xst, res, rst, ws + p.off_res0, L.ln1_w, nred, xs, l > 0);
      else norm_rows<Tag, 4>(p, c, x, xst, res, rst, ws + p.off_res0, L.ln1_w, nred, xs, l > 0);
      stamp(l, 2);
      projection<Tag, true, false>(p, c, p.qkv, L.wqkv, L.bqkv, xs, nullptr, 0, ws + p.off_qkv, p.nqkv, red4, p.npre);
    }
    stamp(l, 3);
    // ---- B: head norms + RoPE + K/V store + attention partials (the o_proj weights start streaming now)
    grid_barrier(c, [&] { issue_prefetch<false>(p.o, L.wo, c, p.npre); });
    stamp(l, 4);
    {
      DecodeParams dp = p.dp;
      const uint16_t *qkv = (const uint16_t *)(ws + p.off_qkv);
      dp.q = qkv, dp.k_new = qkv + p.q_size, dp.v_new = qkv + p.q_size + p.kv_size;
      dp.k_cache = L.k_cache, dp.v_cache = L.v_cache, dp.q_norm_w = L.q_norm_w, dp.k_norm_w = L.k_norm_w;
      dp.out = ws + p.off_attn, dp.ws_acc = (float *)(ws + p.off_wsacc), dp.ws_ml = (float *)(ws + p.off_wsml);
      const int total = p.att_gx * p.att_gy * p.M, slot = c.tid >> 8;
      constexpr int kSlotFloats = DecodeLds<Tag, D, G, WIDE>::ML_FLOATS + DecodeLds<Tag, D, G, WIDE>::ACC_FLOATS;
      for (int base = 0; base < total; base += 2 * c.nwg) {
        const int vi = base + slot * c.nwg + c.wg;
        const bool on = vi < total;
        const int vv = on ? vi : total - 1;
        paged_decode_body<Tag, D, G, 8, true, WIDE, true, true>(dp, vv % p.att_gx, (vv / p.att_gx) % p.att_gy,
                                                                 vv / (p.att_gx * p.att_gy), c.tid & 255, on,
                                                                 att_lds + slot * kSlotFloats);
        __syncthreads();
      }
    }
    stamp(l, 5);
    // ---- C: split-KV combine
    if (p.dp.nsplit > 1) {
      grid_barrier(c, none);
      stamp(l, 6);
      const int total = p.M * p.dp.n_heads, slot = c.tid >> 7;
      for (int base = 0; base < total; base += 4 * c.nwg) {
        const int vi = base + slot * c.nwg + c.wg;
        const bool on = vi < total;
        paged_decode_reduce_body<Tag, D, true>((const float *)(ws + p.off_wsacc), (const float *)(ws + p.off_wsml),
                                               (typename elem<Tag>::T *)(ws + p.off_attn), p.dp.nsplit, on ? vi : total - 1,
                                               c.tid & 127, on, att_lds + slot * ReduceLds<D>::FLOATS);
        __syncthreads();
      }
    }
    // ---- D: o_proj
    stamp(l, 7);
    grid_barrier(c, none);
    stamp(l, 8);
    projection<Tag, false, false>(p, c, p.o, L.wo, nullptr, nullptr, ws + p.off_attn, p.q_size, ws + p.off_o, K, red4, p.npre);
    stamp(l, 9);
    // ---- E: post-attention norm + gate_up + SiluAndMul
    grid_barrier(c, [&] { issue_prefetch<true>(p.gu, L.wgu, c, p.npre); });
    stamp(l, 10);
    if (p.norm_pt <= 1) norm_rows<Tag, 1>(p, c, ws + p.off_o, K, ws + p.off_res0, K, ws + p.off_res1, L.ln2_w, nred, xs, true);
    else if (p.norm_pt == 2) norm_rows<Tag, 2>(p, c, ws + p.off_o, K, ws + p.off_res0, K, ws + p.off_res1, L.ln2_w, nred, xs, true);
    else norm_rows<Tag, 4>(p, c, ws + p.off_o, K, ws + p.off_res0, K, ws + p.off_res1, L.ln2_w, nred, xs, true);
    stamp(l, 11);
    projection<Tag, true, true>(p, c, p.gu, L.wgu, nullptr, xs, nullptr, 0, ws + p.off_act, p.inter, red4, p.npre);
    stamp(l, 12);
    // ---- F: down_proj
    grid_barrier(c, [&] { issue_prefetch<false>(p.down, L.wdown, c, p.npre); });
    stamp(l, 13);
    projection<Tag, false, false>(p, c, p.down, L.wdown, nullptr, nullptr, ws + p.off_act, p.inter, ws + p.off_h, K, red4, p.npre);
    stamp(l, 14);
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) p.status[0] = ld_rlx(c.bar->giveup);
}

// linear_small_m.hip: run<GATED, PACKED>'s launch plan for <= 16 rows on the packed copy: (K slices, block size in k-steps)
void small_m_plan(int m, int n, int k, bool gated, int *nslices, int *kU, int *rpw_out) {
  const int mt = 1, steps = k / 32;
  const int rpw = gated ? 1 : (n > 4096 ? 2 : 1);
  const int na = gated ? 2 * rpw : rpw;
  const int unroll = (na + mt <= 2) ? 8 : ((na + mt <= 5) ? 4 : 2);
  const int nblk = (steps + unroll - 1) / unroll;
  int waves = 7;
  for (int w = gated ? 8 : 16; w >= 6; --w)
    if (nblk % w == 0) {
      waves = w;
      break;
    }
  if (waves > nblk) waves = nblk;
  while (waves > 1 && waves * na * mt > 64) --waves;
  // from 12 rows up that launch stages x through LDS (one block of `unroll` k-steps x 16 rows per wave, 64 KB in all):
  // fewer, longer K slices where the waves' blocks would not fit (down_proj of Qwen3-1.7B: 12 -> 6 slices)
  if (m >= 12 && rpw < 4) {
    const size_t xblock = (size_t)mt * 16 * unroll * 64;
    while (waves > 1 && waves * xblock > 65536) waves /= 2;
  }
  *nslices = waves, *kU = unroll, *rpw_out = rpw;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
  size_t res0, res1, h, qkv, attn, o, act, wsacc, wsml, total;
};
Layout make_layout(int m, int hidden, int nqkv, int q_size, int inter, int n_heads, int head_dim, int nsplit) {
  Layout L;
  size_t off = align256(sizeof(BarState));
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
  L.res0 = take((size_t)m * hidden * 2), L.res1 = take((size_t)m * hidden * 2), L.h = take((size_t)m * hidden * 2);
  L.qkv = take((size_t)m * nqkv * 2), L.attn = take((size_t)m * q_size * 2), L.o = take((size_t)m * hidden * 2);
  L.act = take((size_t)m * inter * 2);
  const size_t slots = (size_t)m * n_heads * (nsplit > 1 ? nsplit : 1);
  L.wsacc = take(slots * head_dim * 4), L.wsml = take(slots * 2 * 4);
  L.total = off + 256;
  return L;
}

int g_cus = 0;
std::once_flag g_cus_once;
int device_cus() {
  std::call_once(g_cus_once, [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_cus = prop.multiProcessorCount;
  });
  return g_cus;
}

}  // namespace

// plan of the attention phase: the decode kernel's own (paged_decode.hip: make_plan), exported for this file
int mv_paged_decode_plan(int batch, int n_heads, int n_kv_heads, int head_dim, int block_size, int max_blocks, int dtype,
                         int flags, int *g_chunk, int *n_chunks, int *nsplit, int *wide, int *mfma, int *uniform);

extern "C" size_t mv_persist_struct_bytes(int which) { return which == 0 ? sizeof(MvPersistLayer) : sizeof(MvPersistArgs); }

extern "C" size_t mv_decode_layers_persistent_workspace_bytes(int m, int hidden, int n_heads, int n_kv_heads, int head_dim,
                                                              int inter, int block_size, int max_blocks) {
  if (m <= 0) return 256;
  int g, nc, ns = 1, wide, mfma, uni;
  if (mv_paged_decode_plan(m, n_heads, n_kv_heads, head_dim, block_size, max_blocks, MV_BF16, 0, &g, &nc, &ns, &wide, &mfma, &uni))
    ns = 64;
  const int q = n_heads * head_dim, kv = n_kv_heads * head_dim;
  return make_layout(m, hidden, q + 2 * kv, q, inter, n_heads, head_dim, ns).total;
}

extern "C" int mv_decode_layers_persistent(const MvPersistLayer *layers_dev, int n_layers, MvPersistArgs *a,
                                           void *workspace, size_t ws_bytes, unsigned *status_dev, int dtype, int flags,
                                           void *stream) {
  if (!layers_dev || !a || !workspace || !status_dev) return MV_E_NULLPTR;
  if (n_layers <= 0 || a->m <= 0) return MV_OK;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  const int m = a->m, hidden = a->hidden, D = a->head_dim, H = a->n_heads, Hkv = a->n_kv_heads, inter = a->inter;
  if (m > 16 || D != 128 || Hkv <= 0 || H % Hkv) return MV_E_SHAPE;
  if (hidden < 256 || (hidden & 63) || hidden > 8192 || inter < 256 || (inter & 63) || (inter & 15)) return MV_E_SHAPE;
  const int q_size = H * D, kv_size = Hkv * D, nqkv = q_size + 2 * kv_size;
  if (!a->x_in || !a->positions || !a->slot_mapping || !a->block_tables || !a->context_lens || !a->cos || !a->sin)
    return MV_E_NULLPTR;
  if (!mv_aligned16(a->x_in) || (a->res_in && !mv_aligned16(a->res_in)) || !mv_aligned16(workspace) ||
      (a->x_in_stride & 7) || (a->res_in_stride & 7))
    return MV_E_ALIGN;
  if (a->max_blocks <= 0 || a->num_blocks <= 0 || a->block_size <= 0 || (a->block_size & 63)) return MV_E_SHAPE;
  int g, nc, ns, wide, mfma, uni;
  if (mv_paged_decode_plan(m, H, Hkv, D, a->block_size, a->max_blocks, dtype, 0, &g, &nc, &ns, &wide, &mfma, &uni))
    return MV_E_SHAPE;
  if (mfma || nc != 1 || g > 2 || !uni) return MV_E_SHAPE;   // vector attention kernel, <= 2 query heads per kv head (register budget)
  Params p{};
  p.layers = layers_dev, p.n_layers = n_layers, p.M = m;
  p.hidden = hidden, p.inter = inter, p.nqkv = nqkv, p.q_size = q_size, p.kv_size = kv_size, p.eps = a->eps;
  int rpw;
  auto plan = [&](Proj &pj, int n, int k, bool gated) {
    pj.n = n, pj.n_tiles = (n + 15) / 16, pj.K = k;
    small_m_plan(m, n, k, gated, &pj.nslices, &pj.kU, &rpw);
    if (rpw != 1 || pj.nslices > kMaxSlices || k / 32 < pj.kU || (k / 32) % pj.kU) return false;
    // every slice is a whole number of kU-step blocks; chunks of 8 steps where every slice allows it, else of 4
    pj.cs = 8;
    const int steps = k / 32, nblk = steps / pj.kU;
    for (int s = 0; s < pj.nslices; ++s) {
      const int len = ((int64_t)nblk * (s + 1) / pj.nslices - (int64_t)nblk * s / pj.nslices) * pj.kU;
      if (len % 8) pj.cs = 4;
      if (len % 4 || len == 0) return false;
    }
    return true;
  };
  if (!plan(p.qkv, nqkv, hidden, false) || !plan(p.o, hidden, q_size, false) || !plan(p.gu, inter, hidden, true) ||
      !plan(p.down, hidden, inter, false))
    return MV_E_SHAPE;
  if (q_size < 256 || (q_size & 63)) return MV_E_SHAPE;
  {  // rmsnorm.hip: launch_rmsnorm's block
    const int nvec = hidden / 8;
    int t = (nvec + 63) / 64 * 64;
    if (t > 256) t = nvec > 1024 ? 1024 : 256;
    const int per = (nvec + t - 1) / t;
    if (t > 256 || per > 4) return MV_E_SHAPE;
    p.norm_threads = t, p.norm_pt = per <= 1 ? 1 : (per == 2 ? 2 : 4);
  }
  const Layout L = make_layout(m, hidden, nqkv, q_size, inter, H, D, ns);
  if (ws_bytes < L.total) return MV_E_WORKSPACE;
  a->out_h_offset = (int64_t)L.h, a->out_res_offset = (int64_t)L.res1;
  p.ws = (char *)workspace;
  p.off_res0 = L.res0, p.off_res1 = L.res1, p.off_h = L.h, p.off_qkv = L.qkv, p.off_attn = L.attn, p.off_o = L.o;
  p.off_act = L.act, p.off_wsacc = L.wsacc, p.off_wsml = L.wsml;
  p.x_in = a->x_in, p.res_in = a->res_in, p.x_in_stride = a->x_in_stride, p.res_in_stride = a->res_in_stride;
  p.status = status_dev;
  DecodeParams &dp = p.dp;
  dp.block_tables = a->block_tables, dp.context_lens = a->context_lens;
  dp.q_stride = nqkv, dp.k_stride = nqkv, dp.v_stride = nqkv, dp.table_stride = a->table_stride, dp.num_blocks = a->num_blocks;
  dp.n_heads = H, dp.n_kv_heads = Hkv, dp.g_total = H / Hkv, dp.block_size = a->block_size;
  dp.bs_shift = (a->block_size & (a->block_size - 1)) == 0 ? __builtin_ctz((unsigned)a->block_size) : -1;
  dp.max_blocks = a->max_blocks, dp.nsplit = ns, dp.scale_log2 = a->scale * 1.4426950408889634f;
  dp.positions = a->positions, dp.cos = a->cos, dp.sin = a->sin, dp.max_pos = a->max_pos, dp.eps = a->eps;
  dp.slot_mapping = a->slot_mapping;
  p.att_gx = wide ? Hkv / 4 : Hkv, p.att_gy = ns;
  // LDS: the weight ring of the 7 streaming waves, then the phase-local scratch (norm sums + normalised rows + partial
  // tiles, or the attention / combine slots)
  int red_tiles = p.qkv.nslices;
  for (int v : {p.o.nslices, 2 * p.gu.nslices, p.down.nslices}) red_tiles = v > red_tiles ? v : red_tiles;
  const size_t proj_scratch = 1024 + (size_t)m * norm_row_bytes(hidden) + (size_t)red_tiles * 1024;
  const int hg = wide ? 4 : 1;
  const size_t att_scratch = 1024 + 2 * (size_t)(4 * hg * g * 2 + 4 * hg * g * D) * 4;   // 2 slots of the decode body
  const size_t scratch = proj_scratch > att_scratch ? proj_scratch : att_scratch;
  p.stamp_layer = (flags & 0x100) ? (n_layers > 1 ? 1 : 0) : -1;   // flags bit 8: phase stamps of layer 1 (diagnostic)
  int npre = (flags & 0xff) == 0xff ? 0 : ((flags & 0xff) ? (flags & 0xff) : 2);    // flags bits 0-7: chunks in flight per wave (0xff = none)
  constexpr size_t kLdsMax = 160 * 1024 - 256;   // (the kernel's 16 static bytes in front)
  while (npre > 0 && (size_t)kStreamWaves * npre * kChunkBytes + scratch > kLdsMax) --npre;
  p.npre = npre;
  const size_t lds = (size_t)kStreamWaves * npre * kChunkBytes + scratch;
  if (lds > kLdsMax) return MV_E_SHAPE;
  const int cus = device_cus();
  if (cus <= 0) return MV_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(zero_state_kernel, dim3(1), dim3(256), 0, st, (unsigned *)workspace, (int)(align256(sizeof(BarState)) / 4));
#define MV_PERSIST(TAG, GG, WW)                                                                                     \
  do {                                                                                                              \
    auto kern = decode_layers_persistent_kernel<TAG, 128, GG, WW>;                                                  \
    static std::once_flag once;                                                                                     \
    static hipError_t attr_rc = hipSuccess;                                                                          \
    std::call_once(once, [&] { attr_rc = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax); }); \
    if (attr_rc != hipSuccess) return (int)attr_rc;                                                                  \
    hipLaunchKernelGGL(kern, dim3(cus), dim3(kThreads), lds, st, p);                                                \
  } while (0)
#define MV_PERSIST_G(TAG)                                          \
  do {                                                             \
    if (wide) {                                                    \
      if (g == 1) MV_PERSIST(TAG, 1, true);                        \
      else if (g == 2) MV_PERSIST(TAG, 2, true);                   \
      else return MV_E_SHAPE;                                      \
    } else {                                                       \
      if (g == 1) MV_PERSIST(TAG, 1, false);                       \
      else if (g == 2) MV_PERSIST(TAG, 2, false);                  \
      else return MV_E_SHAPE;                                      \
    }                                                              \
  } while (0)
  if (dtype == MV_BF16) MV_PERSIST_G(bf16_tag);
  else MV_PERSIST_G(f16_tag);
#undef MV_PERSIST_G
#undef MV_PERSIST
  MV_LAUNCH_CHECK();
  return MV_OK;
}
