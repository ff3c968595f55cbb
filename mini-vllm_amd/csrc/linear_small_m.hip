// K10 — y[M,N] = x[M,K] · W[N,K]ᵀ (+ bias) for the decode-sized M (≤ 64 rows): the weight-streaming
// form of the reference's F.linear calls (minivllm/models/layers/linear.py:130,163,316,
// layers/embed_head.py:217) when a decode step carries 1..64 tokens.
//
// At these M every projection is a pure HBM stream of W (2·M flop per weight byte ≪ the machine
// balance), and the library GEMMs are latency-bound on their few output tiles (11–17 µs for 8–50 MB
// of weights).  Here a workgroup owns RPW·16 weight rows and ALL of K:
//   * each wave takes a contiguous K slice of those rows and streams it straight from HBM into
//     the A operand of v_mfma_f32_16x16x32 (lane = (row, 8-element k group): 16 rows × 64 B per
//     load instruction, every load of the slice in flight before the first wait) — no LDS on the
//     weight path (guide §5, "GEMV / M ≤ 16" row);
//   * x (L2-resident, ≤ 0.8 MB) is read in the same fragment shape as the B operand, MT = ⌈M/16⌉
//     fragments per k step; rows beyond M re-read row M-1 and their output columns are dropped;
//   * the waves' partial 16×16 tiles are summed through LDS in wave order (deterministic),
//     bias is added in fp32, one rounding to the output dtype (the GEMM epilogue's rounding point).
// GATED variant (mv_gate_up_silu_small_m): W = [gate rows | up rows] of the merged gate_up projection; a
// wave streams RPW gate tiles AND the RPW up tiles of the same output columns, and the epilogue
// applies SiluAndMul with the rounding points of the unfused pair (GEMM output -> dtype, silu -> dtype,
// product -> dtype), so the activation never exists as a [M, 2I] tensor and one launch disappears.
// No workspace, no atomics, no inter-workgroup traffic: hipGraph-capturable as is.
#include "common.h"

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

struct LinearParams {
  const void *x, *w, *bias;
  void *out;
  int64_t x_stride, out_stride;
  int M, N, K;  // N = output columns (GATED: W has 2N rows, up rows start at row N)
  // NORM: x is the input of the RMSNorm in front of the projection (see norm_prologue)
  const void *nres = nullptr, *nw = nullptr;
  void *nres_out = nullptr;
  int64_t nres_stride = 0, nres_out_stride = 0;
  float eps = 0.f;
  int norm_threads = 0;  // block size of the rmsnorm_kernel launch this prologue re-enacts
};

constexpr int kNormRows = 8;        // rows the norm prologue takes (round 4: 8; rounds 2-3: 4)
constexpr int kNormRedBytes = kNormRows * 16 * 4;  // [8 rows][16 waves] floats in front of the normalised rows
__host__ __device__ constexpr int norm_row_bytes(int k) { return k * 2 + 64; }  // (+64: rows 0..3 on disjoint banks)

// NORM prologue: the residual-add + RMSNorm that the reference runs in front of this projection
// (minivllm/models/layers/layernorm.py:81-105; rmsnorm.hip) re-enacted by EVERY workgroup for the <= 8 rows of a
// small decode step — a few KB from L2 — so that the norm needs no launch of its own.  The arithmetic is
// rmsnorm_kernel's, thread for thread: a block of `norm_threads` threads, thread t owning the 8-element vectors
// t, t + T, ..., the same fmaf chain, wave_sum, and wave-order sum, the same rounding points (the norm is taken
// of the unrounded fp32 sum x + residual; the rounded sum is the new residual, written by workgroup 0 only and
// to a buffer of its own: the other workgroups still read the old one).  The normalised rows land in LDS.
template <typename Tag, int PT>
__device__ __forceinline__ void norm_prologue(const LinearParams &p, float *nred, char *xs) {
  constexpr int ROWS = 4 / PT;   // rows per pass: their vectors live in registers between the two halves of the pass
  const int T = p.norm_threads, nvec = p.K >> 3, tid = threadIdx.x, nw = T >> 6;
  const uint16_t *x = (const uint16_t *)p.x, *res = (const uint16_t *)p.nres, *w = (const uint16_t *)p.nw;
  uint16_t *res_out = (uint16_t *)p.nres_out;
  const bool add = res != nullptr;
  u32x4 wraw[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int c = tid + i * T;
    wraw[i] = (tid < T && c < nvec) ? *(const u32x4 *)(w + c * 8) : u32x4{0u, 0u, 0u, 0u};
  }
  // round 4: more than ROWS rows (up to kNormRows) go through in passes of ROWS; a pass only touches its own rows' sums, so one
  // barrier per pass (between the sums and their use) is enough
  for (int r0 = 0; r0 < p.M; r0 += ROWS) {
    u32x4 xraw[ROWS][PT], rraw[ROWS][PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int c = tid + i * T;
      const bool live = tid < T && c < nvec;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const bool on = live && r0 + r < p.M;
        xraw[r][i] = on ? *(const u32x4 *)(x + (int64_t)(r0 + r) * p.x_stride + c * 8) : u32x4{0u, 0u, 0u, 0u};
        rraw[r][i] = on && add ? *(const u32x4 *)(res + (int64_t)(r0 + r) * p.nres_stride + c * 8) : u32x4{0u, 0u, 0u, 0u};
      }
    }
    float vals[ROWS][PT][8];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < PT; ++i) {
        const int c = tid + i * T;
        const bool on = tid < T && c < nvec && r0 + r < p.M;
        unpack8<Tag>(xraw[r][i], vals[r][i]);
        if (add) {
          float rr[8];
          unpack8<Tag>(rraw[r][i], rr);
#pragma unroll
          for (int j = 0; j < 8; ++j) vals[r][i][j] = __fadd_rn(vals[r][i][j], rr[j]);
          if (on && blockIdx.x == 0) *(u32x4 *)(res_out + (int64_t)(r0 + r) * p.nres_out_stride + c * 8) = pack8<Tag>(vals[r][i]);
        }
        if (on) {
#pragma unroll
          for (int j = 0; j < 8; ++j) ss = fmaf(vals[r][i][j], vals[r][i][j], ss);
        }
      }
      ss = wave_sum(ss);
      if ((tid & 63) == 0 && (tid >> 6) < nw && r0 + r < p.M) nred[(r0 + r) * 16 + (tid >> 6)] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (r0 + r < p.M && tid < T) {
        float t = nred[(r0 + r) * 16];
        for (int i = 1; i < nw; ++i) t += nred[(r0 + r) * 16 + i];   // block_sum: 0.f + red[0] + red[1] + ...
        const float var = t / (float)p.K;
        const float rs = 1.0f / sqrtf(var + p.eps);
#pragma unroll
        for (int i = 0; i < PT; ++i) {
          const int c = tid + i * T;
          if (c < nvec) {
            float wv[8], y[8];
            unpack8<Tag>(wraw[i], wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = __fmul_rn(__fmul_rn(vals[r][i][j], rs), wv[j]);
            *(u32x4 *)(xs + (r0 + r) * norm_row_bytes(p.K) + c * 16) = pack8<Tag>(y);
          }
        }
      }
    }
  }
  __syncthreads();
}


// k steps (32 elements each) whose loads are issued together before the first MFMA
// (16–20 wave-wide loads in flight per wave)
__host__ __device__ constexpr int unroll_of(int rpw, int mt) { return rpw + mt <= 2 ? 8 : (rpw + mt <= 5 ? 4 : 2); }

// XLDS: the x block of a wave (MT·16 rows x kUnroll·32 elements) comes in by LDS-DMA in whole-row pieces
// (2-4 rows x 256-512 B per instruction, XOR swizzle applied on the source address) and the B fragments are
// ds_read_b128 -- instead of 16 rows x 64 B per load instruction straight into the B-operand registers, which
// costs the texture path as much as the weight fragments do.  Chosen by run() where it measured faster.
// WLDS: the same for the weight block of a wave that owns one 16-row tile (x then stays in registers).
// PACKED: W is the fragment-ordered copy mv_pack_weight makes for the decode-batch kernel ([N/16][K/32][64 lanes][8]:
// the A fragment of a (16-row tile, k step) is 1 KiB of contiguous memory and a wave's K slice of a tile one
// contiguous run), read non-temporally: every load instruction is a whole-line burst instead of 16 pieces of 64 B.
template <typename Tag, int RPW, int MT, bool GATED, bool XLDS, bool WLDS = false, bool PACKED = false, bool NORM = false>
__global__ void __launch_bounds__((GATED ? 2 : 1) * RPW * MT >= 12 ? 512 : 1024)
linear_small_m_kernel(LinearParams p) {
  static_assert(!NORM || (PACKED && MT == 1 && !XLDS), "the norm prologue exists for the packed, <= 16-row form");  // (>= 12 tiles per wave: the 64 KB of partial-tile LDS allow <= 5 waves)
  typedef uint16_t T;
  extern __shared__ float red[];  // [waves][RPW*MT][64][4]; XLDS: first the per-wave x blocks
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * (RPW * 16);
  constexpr int NA = GATED ? 2 * RPW : RPW;  // weight tiles per wave
  constexpr int kUnroll = unroll_of(NA, MT);
  const int steps = p.K >> 5;
  int s_lo, s_hi;
  if constexpr (PACKED) {
    // K slices in whole blocks of kUnroll steps: no wave but the last re-reads part of a block
    const int nblk = (steps + kUnroll - 1) / kUnroll;
    s_lo = (int)((int64_t)nblk * wave / nwaves) * kUnroll;
    s_hi = min(steps, (int)((int64_t)nblk * (wave + 1) / nwaves) * kUnroll);
  } else {
    s_lo = (int)((int64_t)steps * wave / nwaves);
    s_hi = (int)((int64_t)steps * (wave + 1) / nwaves);
  }

  static_assert(!(PACKED && WLDS), "packed weights go straight to registers");
  constexpr int WSTEP = PACKED ? 512 : 32;  // elements between the fragments of consecutive k steps
  const T *wp[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    if constexpr (PACKED) {
      // tile index: the packed copy pads N up to whole tiles (zero rows), GATED: the up rows start at tile N / 16
      const int t = blockIdx.x * RPW + (a % RPW) + (GATED && a >= RPW ? p.N >> 4 : 0);
      const int t_last = ((GATED ? 2 * p.N : p.N) + 15) / 16 - 1;
      wp[a] = (const T *)p.w + (int64_t)(t < t_last ? t : t_last) * steps * 512 + lane * 8;
    } else {
      int n = n0 + (a % RPW) * 16 + r16;
      n = n < p.N ? n : p.N - 1;  // ragged last tile: re-read the last row, dropped at the store
      if (GATED && a >= RPW) n += p.N;
      wp[a] = (const T *)p.w + (int64_t)n * p.K + g * 8;
    }
  }
  const T *xp[MT];
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    int m = b * 16 + r16;
    m = m < p.M ? m : p.M - 1;
    xp[b] = (const T *)p.x + (int64_t)m * p.x_stride + g * 8;
  }

  f32x4 acc[NA][MT];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char *xs_row = nullptr;  // NORM: this lane's normalised row in LDS (+ its k-group offset)
  // NORM: the first weight block of this wave's K slice is requested BEFORE the norm prologue — the weights do
  // not depend on it, and the prologue's two L2 round trips then hide behind the HBM latency of the stream
  u32x4 wf0[NORM ? kUnroll : 1][NORM ? NA : 1];
  if constexpr (NORM) {
    if (s_lo < s_hi) {
      int sb = s_lo + kUnroll <= s_hi ? s_lo : s_hi - kUnroll;
      sb = sb < 0 ? 0 : sb;
#pragma unroll
      for (int u = 0; u < kUnroll; ++u)
#pragma unroll
        for (int a = 0; a < NA; ++a)
          wf0[u][a] = __builtin_nontemporal_load((const u32x4 *)(wp[a] + (int64_t)(sb + u) * WSTEP));
    }
    char *xs = (char *)red + kNormRedBytes;
    const int pt = ((p.K >> 3) + p.norm_threads - 1) / p.norm_threads;
    if (pt <= 1) norm_prologue<Tag, 1>(p, red, xs);
    else if (pt == 2) norm_prologue<Tag, 2>(p, red, xs);
    else norm_prologue<Tag, 4>(p, red, xs);
    xs_row = xs + (r16 < p.M ? r16 : p.M - 1) * norm_row_bytes(p.K) + g * 16;
  }

  // blocks of kUnroll k-steps, all loads of a block issued before its first MFMA; the last block of a
  // slice is shifted back to end at the slice's end and feeds zeros for the steps already done (no
  // serial tail; the host guarantees K/32 >= kUnroll so the shifted window stays inside the row)
  constexpr int RB = kUnroll * 64;             // bytes of one x row in a block
  constexpr int XBLOCK = (WLDS ? NA : MT) * 16 * RB;  // bytes of a wave's x (XLDS) or weight (WLDS) block
  constexpr int XM = (RB / 16 - 1) & 15;       // swizzle mask: stays inside the row's chunks
  char *xl = (char *)red + wave * XBLOCK;      // XLDS
  for (int s = s_lo; s < s_hi; s += kUnroll) {
    int sb = s + kUnroll <= s_hi ? s : s_hi - kUnroll;
    sb = sb < 0 ? 0 : sb;
    u32x4 wf[kUnroll][NA], xf[kUnroll][MT];
    if constexpr (WLDS) {
      static_assert(!WLDS || !XLDS, "W through LDS: x goes straight to registers");
      // weight block: NA tiles x 16 rows x RB bytes by whole-row DMA (swizzled on the source), A fragments by
      // ds_read_b128
#pragma unroll
      for (int n = 0; n < (NA * 16 * RB) / 1024; ++n) {
        const int o = n * 1024 + lane * 16;
        const int trow = o / RB, pc = (o % RB) >> 4;   // trow = tile * 16 + row
        const int a = trow >> 4, row = trow & 15;
        int nrow = n0 + (a % RPW) * 16 + row;
        nrow = nrow < p.N ? nrow : p.N - 1;
        if (GATED && a >= RPW) nrow += p.N;
        const T *src = (const T *)p.w + (int64_t)nrow * p.K + sb * 32 + ((pc ^ (row & XM)) << 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(xl + n * 1024), 16, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      if constexpr (!WLDS) {
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          if constexpr (NORM) {
            if (s == s_lo) wf[u][a] = wf0[u][a];   // (wave-uniform)
            else wf[u][a] = __builtin_nontemporal_load((const u32x4 *)(wp[a] + (int64_t)(sb + u) * WSTEP));
          } else if constexpr (PACKED)
            wf[u][a] = __builtin_nontemporal_load((const u32x4 *)(wp[a] + (int64_t)(sb + u) * WSTEP));
          else
            wf[u][a] = *(const u32x4 *)(wp[a] + sb * 32 + u * 32);
        }
      }
      if constexpr (NORM) {
        xf[u][0] = *(const u32x4 *)(xs_row + (sb + u) * 64);
      } else if constexpr (!XLDS) {
#pragma unroll
        for (int b = 0; b < MT; ++b) xf[u][b] = *(const u32x4 *)(xp[b] + sb * 32 + u * 32);
      }
    }
    if constexpr (XLDS) {
      // instruction n fills LDS bytes [1024 n, 1024 (n+1)) of the block linearly: lane -> (row, stored chunk);
      // it fetches the LOGICAL chunk (stored ^ (row & 15)) of that row, so that the B-fragment reads below
      // (16 rows, one logical chunk) hit 16 different bank groups
#pragma unroll
      for (int n = 0; n < XBLOCK / 1024; ++n) {
        const int o = n * 1024 + lane * 16;
        const int row = o / RB, pc = (o % RB) >> 4;
        const int m = row < p.M ? row : p.M - 1;
        const T *src = (const T *)p.x + (int64_t)m * p.x_stride + sb * 32 + ((pc ^ (row & XM)) << 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(xl + n * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA is invisible to the compiler's own counting
#pragma unroll
      for (int u = 0; u < kUnroll; ++u)
#pragma unroll
        for (int b = 0; b < MT; ++b) {
          const int row = b * 16 + r16;
          xf[u][b] = *(const u32x4 *)(xl + row * RB + (((4 * u + g) ^ (row & XM)) << 4));
        }
    }
    if constexpr (WLDS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < kUnroll; ++u)
#pragma unroll
        for (int a = 0; a < NA; ++a)
          wf[u][a] = *(const u32x4 *)(xl + (a * 16 + r16) * RB + (((4 * u + g) ^ (r16 & XM)) << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      if (sb + u < s || sb + u >= s_hi) {
#pragma unroll
        for (int b = 0; b < MT; ++b) xf[u][b] = u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = mfma16<Tag>(wf[u][a], xf[u][b], acc[a][b]);
    }
  }

  // partial tiles -> LDS, then output tile t of the workgroup is finished by wave t % nwaves
  constexpr int NT = NA * MT;
  f32x4 *red4 = (f32x4 *)red;
  if constexpr (XLDS || WLDS || NORM) __syncthreads();  // the partial tiles overwrite the blocks / rows other waves read
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) red4[(wave * NT + a * MT + b) * 64 + lane] = acc[a][b];
  __syncthreads();
  auto tile_sum = [&](int t) {
    f32x4 sum = red4[t * 64 + lane];
    for (int w = 1; w < nwaves; ++w) {
      const f32x4 v = red4[(w * NT + t) * 64 + lane];
      sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
    }
    return sum;
  };
  for (int t = wave; t < RPW * MT; t += nwaves) {
    const int a = t / MT, b = t % MT;
    // C layout: lane holds W rows n = g*4 + {0..3} (consecutive outputs) of token m = r16
    const int m = b * 16 + r16, n = n0 + a * 16 + g * 4;
    const f32x4 sum = tile_sum(t);
    float o[4] = {sum[0], sum[1], sum[2], sum[3]};
    if constexpr (GATED) {
      const f32x4 up = tile_sum(t + RPW * MT);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float gt = elem<Tag>::ld(elem<Tag>::st(o[i]));  // the GEMM output as the unfused path stores it
        const float ut = elem<Tag>::ld(elem<Tag>::st(up[i]));
        float sl = gt / (1.0f + expf(-gt));                     // rmsnorm.hip: silu_and_mul_kernel
        sl = elem<Tag>::ld(elem<Tag>::st(sl));
        o[i] = __fmul_rn(sl, ut);
      }
    }
    if (m < p.M && n < p.N) {
      if (!GATED && p.bias) {
        const T *bp = (const T *)p.bias + n;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] += elem<Tag>::ld(bp[i]);
      }
      u32x2 pk;
      pk[0] = (uint32_t)elem<Tag>::st(o[0]) | ((uint32_t)elem<Tag>::st(o[1]) << 16);
      pk[1] = (uint32_t)elem<Tag>::st(o[2]) | ((uint32_t)elem<Tag>::st(o[3]) << 16);
      *(u32x2 *)((T *)p.out + (int64_t)m * p.out_stride + n) = pk;
    }
  }
}

template <typename Tag, int RPW, bool GATED, bool XLDS, bool PACKED>
int launch_mt_(const LinearParams &p, int mt, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
  switch (mt) {
    case 1: hipLaunchKernelGGL((linear_small_m_kernel<Tag, RPW, 1, GATED, XLDS, false, PACKED>), grid, block, lds, stream, p); break;
    case 2: hipLaunchKernelGGL((linear_small_m_kernel<Tag, RPW, 2, GATED, XLDS, false, PACKED>), grid, block, lds, stream, p); break;
    case 3: hipLaunchKernelGGL((linear_small_m_kernel<Tag, RPW, 3, GATED, XLDS, false, PACKED>), grid, block, lds, stream, p); break;
    default: hipLaunchKernelGGL((linear_small_m_kernel<Tag, RPW, 4, GATED, XLDS, false, PACKED>), grid, block, lds, stream, p); break;
  }
  MV_LAUNCH_CHECK();
  return MV_OK;
}

template <typename Tag, bool GATED, bool XLDS, bool PACKED>
int launch(const LinearParams &p, int rpw, int mt, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
  if constexpr (!GATED)
    if (rpw == 4) return launch_mt_<Tag, 4, GATED, XLDS, PACKED>(p, mt, grid, block, lds, stream);
  if (rpw == 2) return launch_mt_<Tag, 2, GATED, XLDS, PACKED>(p, mt, grid, block, lds, stream);
  return launch_mt_<Tag, 1, GATED, XLDS, PACKED>(p, mt, grid, block, lds, stream);
}

// `shape` (the last-but-one argument of the entry points) overrides the launch shape, for tools/bench_linear.py
// --sweep: bits 0-3 rows-per-wave tiles (1, 2, 4), bits 4-8 waves, bits 9-10 x through LDS (1 on, 2 off),
// bits 11-12 W through LDS (1 on, 2 off); 0 = everything chosen from the shape

// n = output columns.  Rows per wave (RPW·16; GATED: RPW gate tiles + RPW up tiles): every workgroup
// re-reads all of x from L2 in fragment-shaped pieces that cost the texture path as much as the weight
// fragments do (free for M <= 2, where the 16 lanes of a row group read the same line), so wide layers
// take 2-4 row tiles per wave.  K slices per workgroup: enough waves that the chip holds >= ~2048 of
// them, each slice at least one full block of k-steps (two for the last doubling).  Measured choices:
// tools/bench_linear.py --sweep.
// the block size launch_rmsnorm (rmsnorm.hip) gives a row of `cols` 16-bit elements, and 16-byte vectors per thread
static void norm_block(int cols, int *threads, int *per_thread) {
  const int nvec = cols / 8;
  int t = (nvec + 63) / 64 * 64;
  if (t > 256) t = nvec > 1024 ? 1024 : 256;
  *threads = t;
  *per_thread = (nvec + t - 1) / t;
}

template <bool GATED, bool PACKED = false, bool NORM = false>
int run(LinearParams p, int dtype, int shape, hipStream_t st) {
  const int m = p.M, n = p.N, mt = (m + 15) / 16, steps = p.K / 32;
  int rpw = GATED ? (m > 2 && n >= 16384 ? 2 : 1) : (m <= 2 ? 1 : (n >= 65536 ? 4 : (n >= 8192 ? 2 : 1)));
  // (on the packed copy 4 row tiles per wave never measured fastest, nor 2 + 2 tiles for the gated form:
  //  Qwen2-7B gate_up at 16 rows 44 us with 1 + 1 tiles over 7 waves, 64 us with 2 + 2)
  if (PACKED) rpw = GATED ? 1 : (n > 4096 ? 2 : 1);
  const int force_rpw = shape & 15;
  if (force_rpw == 1 || force_rpw == 2 || (force_rpw == 4 && !GATED)) rpw = force_rpw;
  const int na = GATED ? 2 * rpw : rpw;
  const int tiles = (n + rpw * 16 - 1) / (rpw * 16);
  const int unroll = unroll_of(na, mt);
  int waves = 4;
  while (waves < 16 && tiles * waves < 2048 && waves * 2 * na * mt <= 64 &&
         steps / (waves * 2) >= (waves == 8 ? 2 * unroll : unroll))
    waves *= 2;
  if (PACKED) {
    // whole-line weight loads (tools/bench_linear.py --packed, M = 1 and 8): what matters is that every wave of the
    // workgroup walks the same number of whole K blocks — the largest wave count in 6..16 that divides them, 7
    // where none does (Qwen2-7B, K = 3584: 7 / 14 waves 8.3 us for the qkv matrix, 8 waves 9.1, 16 waves 9.0)
    const int nblk = (steps + unroll - 1) / unroll;
    // (the gated form carries two tiles per wave: 6..8 waves; 14-16 measured 1-15 % slower at 8-16 rows)
    waves = 7;
    for (int w = GATED ? 8 : 16; w >= 6; --w)
      if (nblk % w == 0) {
        waves = w;
        break;
      }
    if (waves > nblk) waves = nblk;
    while (waves > 1 && waves * na * mt > 64) --waves;
  }
  const int force_waves = (shape >> 4) & 31;
  if (force_waves >= 1 && force_waves <= 16 && force_waves * na * mt <= 64) waves = force_waves;
  while (waves > 1 && waves * na * mt > 64) waves /= 2;  // 64 KB of LDS for the partial tiles
  if constexpr (NORM) {
    int nt, npt;
    norm_block(p.K, &nt, &npt);
    const int pt = npt <= 1 ? 1 : (npt == 2 ? 2 : 4);
    if (mt != 1 || npt > 4 || m * pt > kNormRows || waves * 64 < nt || (p.K & 7)) return MV_E_SHAPE;
    p.norm_threads = nt;
    size_t lds = (size_t)waves * na * 64 * sizeof(f32x4);
    const size_t need = kNormRedBytes + (size_t)m * norm_row_bytes(p.K);
    if (need > lds) lds = need;
    if (lds > 65536) return MV_E_SHAPE;
    dim3 grid(tiles), block(waves * 64);
#define MV_NORM(TAG)                                                                                                  \
  do {                                                                                                                \
    if (rpw == 2 && !GATED)                                                                                           \
      hipLaunchKernelGGL((linear_small_m_kernel<TAG, GATED ? 1 : 2, 1, GATED, false, false, true, true>), grid, block, lds, st, p); \
    else                                                                                                              \
      hipLaunchKernelGGL((linear_small_m_kernel<TAG, 1, 1, GATED, false, false, true, true>), grid, block, lds, st, p); \
  } while (0)
    if (rpw != 1 && rpw != 2) return MV_E_SHAPE;
    if (dtype == MV_BF16) MV_NORM(bf16_tag);
    else MV_NORM(f16_tag);
#undef MV_NORM
    MV_LAUNCH_CHECK();
    return MV_OK;
  }
  // x through LDS once its fragment-shaped loads weigh: from 12 rows up, unless 4 row tiles per wave already
  // amortise them (measured, tools/dbg/xlds_bench.sh: M=16 +5..17 % on the 8-34 MB layers, -15 % on the LM
  // head with RPW 4; M=32 +7..30 % everywhere; M <= 8 -1..-10 %)
  const int force_xlds = ((shape >> 9) & 3) == 1 ? 1 : (((shape >> 9) & 3) == 2 ? -1 : 0);
  const bool xlds = force_xlds ? force_xlds > 0 : (mt >= 2 || (m >= 12 && rpw < 4));
  const size_t xblock = (size_t)mt * 16 * unroll * 64;
  while (xlds && waves > 1 && waves * xblock > 65536) waves /= 2;
  size_t lds = (size_t)waves * na * mt * 64 * sizeof(f32x4);
  if (xlds && waves * xblock > lds) lds = waves * xblock;
  dim3 grid(tiles), block(waves * 64);
  // one or two weight tiles per wave and more than one block per K slice (K >= 3584, or 6144 over 8 waves): the
  // weight block also comes in by whole-row DMA (+13..25 % on the down / Qwen2-7B qkv projections at M <= 8,
  // -5 % on single-block slices and on the LM head, tools/dbg/wlds_bench.sh); MV_LINEAR_WLDS=1/-1 forces
  const int force_wlds = ((shape >> 11) & 3) == 1 ? 1 : (((shape >> 11) & 3) == 2 ? -1 : 0);
  const bool wlds = !PACKED && (force_wlds ? force_wlds > 0 : (steps / waves > unroll && n < 65536));
  if (wlds && !xlds && mt == 1 && na <= 2) {
    const size_t wblock = (size_t)na * 16 * unroll * 64;
    while (waves > 1 && waves * wblock > 65536) waves /= 2;
    if (waves * wblock > lds) lds = waves * wblock;
    dim3 blk(waves * 64);
#define MV_WLDS(TAG)                                                                                             \
  do {                                                                                                           \
    if (rpw == 2) hipLaunchKernelGGL((linear_small_m_kernel<TAG, 2, 1, GATED, false, true>), grid, blk, lds, st, p); \
    else hipLaunchKernelGGL((linear_small_m_kernel<TAG, 1, 1, GATED, false, true>), grid, blk, lds, st, p);      \
  } while (0)
    if constexpr (GATED) {
      if (rpw == 1) {
        if (dtype == MV_BF16) hipLaunchKernelGGL((linear_small_m_kernel<bf16_tag, 1, 1, true, false, true>), grid, blk, lds, st, p);
        else hipLaunchKernelGGL((linear_small_m_kernel<f16_tag, 1, 1, true, false, true>), grid, blk, lds, st, p);
        MV_LAUNCH_CHECK();
        return MV_OK;
      }
    } else {
      if (dtype == MV_BF16) MV_WLDS(bf16_tag);
      else MV_WLDS(f16_tag);
      MV_LAUNCH_CHECK();
      return MV_OK;
    }
#undef MV_WLDS
  }
  if (xlds)
    return dtype == MV_BF16 ? launch<bf16_tag, GATED, true, PACKED>(p, rpw, mt, grid, block, lds, st)
                            : launch<f16_tag, GATED, true, PACKED>(p, rpw, mt, grid, block, lds, st);
  return dtype == MV_BF16 ? launch<bf16_tag, GATED, false, PACKED>(p, rpw, mt, grid, block, lds, st)
                          : launch<f16_tag, GATED, false, PACKED>(p, rpw, mt, grid, block, lds, st);
}

int check(const void *x, int64_t x_stride, const void *w, const void *bias, void *out, int64_t out_stride, int m,
          int n, int k, int dtype) {
  if (!x || !w || !out) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (m < 0 || m > 64 || n < 0 || k < 256 || (k & 31) || (n & 3)) return MV_E_SHAPE;
  if (!mv_aligned16(x) || !mv_aligned16(w) || (((uintptr_t)out) & 7u) || (x_stride & 7) || (out_stride & 3) ||
      (bias && (((uintptr_t)bias) & 7u)))
    return MV_E_ALIGN;
  return MV_OK;
}

}  // namespace

extern "C" int mv_linear_small_m(const void *x, int64_t x_stride, const void *w, const void *bias, void *out,
                                 int64_t out_stride, int m, int n, int k, int dtype, int shape, void *stream) {
  if (m == 0 || n == 0) return MV_OK;
  if (int e = check(x, x_stride, w, bias, out, out_stride, m, n, k, dtype)) return e;
  LinearParams p{x, w, bias, out, x_stride, out_stride, m, n, k};
  return run<false>(p, dtype, shape, (hipStream_t)stream);
}

extern "C" int mv_gate_up_silu_small_m(const void *x, int64_t x_stride, const void *w_gate_up, void *out,
                                       int64_t out_stride, int m, int n_inter, int k, int dtype, int shape,
                                       void *stream) {
  if (m == 0 || n_inter == 0) return MV_OK;
  if (int e = check(x, x_stride, w_gate_up, nullptr, out, out_stride, m, n_inter, k, dtype)) return e;
  LinearParams p{x, w_gate_up, nullptr, out, x_stride, out_stride, m, n_inter, k};
  return run<true>(p, dtype, shape, (hipStream_t)stream);
}

// The same two entry points on the fragment-ordered weight copy of mv_pack_weight (include/mvhip.h): same
// arithmetic, same summation order and rounding points — only where the weight bytes come from differs.
extern "C" int mv_linear_small_m_packed(const void *x, int64_t x_stride, const void *w_packed, const void *bias,
                                        void *out, int64_t out_stride, int m, int n, int k, int dtype, int shape,
                                        void *stream) {
  if (m == 0 || n == 0) return MV_OK;
  if (int e = check(x, x_stride, w_packed, bias, out, out_stride, m, n, k, dtype)) return e;
  LinearParams p{x, w_packed, bias, out, x_stride, out_stride, m, n, k};
  return run<false, true>(p, dtype, shape, (hipStream_t)stream);
}

extern "C" int mv_gate_up_silu_small_m_packed(const void *x, int64_t x_stride, const void *w_gate_up_packed,
                                              void *out, int64_t out_stride, int m, int n_inter, int k, int dtype,
                                              int shape, void *stream) {
  if (m == 0 || n_inter == 0) return MV_OK;
  if (n_inter & 15) return MV_E_SHAPE;  // the up rows must start on a tile boundary of the packed copy
  if (int e = check(x, x_stride, w_gate_up_packed, nullptr, out, out_stride, m, n_inter, k, dtype)) return e;
  LinearParams p{x, w_gate_up_packed, nullptr, out, x_stride, out_stride, m, n_inter, k};
  return run<true, true>(p, dtype, shape, (hipStream_t)stream);
}

// The RMSNorm in front of a projection, folded into the projection's launch (<= 4 rows; see norm_prologue):
//   residual == NULL:  out = Linear(RMSNorm(x))                                   (the first layer's input norm)
//   else:              residual_out = x + residual;  out = Linear(RMSNorm(x + residual))   (add_rms_forward)
// with every value bit-identical to mv_rmsnorm / mv_add_rmsnorm followed by mv_linear_small_m_packed.
// MV_E_SHAPE when the fused form does not take the shape: the caller then launches the two kernels.
static int norm_check(const void *x, int64_t x_stride, const void *residual, int64_t res_stride, void *residual_out,
                      int64_t res_out_stride, const void *norm_w, int m, int k) {
  if (!norm_w || (residual && !residual_out)) return MV_E_NULLPTR;
  if (m > kNormRows || k > 8192) return MV_E_SHAPE;
  if (!mv_aligned16(norm_w) || (residual && (!mv_aligned16(residual) || !mv_aligned16(residual_out) ||
                                             (res_stride & 7) || (res_out_stride & 7))))
    return MV_E_ALIGN;
  if (residual && residual_out == residual) return MV_E_ALIAS;  // the other workgroups still read the old rows
  return MV_OK;
}

extern "C" int mv_norm_linear_small_m_packed(const void *x, int64_t x_stride, const void *residual,
                                             int64_t res_stride, void *residual_out, int64_t res_out_stride,
                                             const void *norm_w, float eps, const void *w_packed, const void *bias,
                                             void *out, int64_t out_stride, int m, int n, int k, int dtype,
                                             int shape, void *stream) {
  if (m == 0 || n == 0) return MV_OK;
  if (int e = check(x, x_stride, w_packed, bias, out, out_stride, m, n, k, dtype)) return e;
  if (int e = norm_check(x, x_stride, residual, res_stride, residual_out, res_out_stride, norm_w, m, k)) return e;
  LinearParams p{x, w_packed, bias, out, x_stride, out_stride, m, n, k};
  p.nres = residual, p.nw = norm_w, p.nres_out = residual_out;
  p.nres_stride = res_stride, p.nres_out_stride = res_out_stride, p.eps = eps;
  return run<false, true, true>(p, dtype, shape, (hipStream_t)stream);
}

extern "C" int mv_norm_gate_up_silu_small_m_packed(const void *x, int64_t x_stride, const void *residual,
                                                   int64_t res_stride, void *residual_out, int64_t res_out_stride,
                                                   const void *norm_w, float eps, const void *w_gate_up_packed,
                                                   void *out, int64_t out_stride, int m, int n_inter, int k,
                                                   int dtype, int shape, void *stream) {
  if (m == 0 || n_inter == 0) return MV_OK;
  if (n_inter & 15) return MV_E_SHAPE;
  if (int e = check(x, x_stride, w_gate_up_packed, nullptr, out, out_stride, m, n_inter, k, dtype)) return e;
  if (int e = norm_check(x, x_stride, residual, res_stride, residual_out, res_out_stride, norm_w, m, k)) return e;
  LinearParams p{x, w_gate_up_packed, nullptr, out, x_stride, out_stride, m, n_inter, k};
  p.nres = residual, p.nw = norm_w, p.nres_out = residual_out;
  p.nres_stride = res_stride, p.nres_out_stride = res_out_stride, p.eps = eps;
  return run<true, true, true>(p, dtype, shape, (hipStream_t)stream);
}
