// C1 (SURVEY §8 b: mv_allreduce_*) — small-message all-reduce over xGMI peer-to-peer loads, for the
// [tokens, hidden] sums after o_proj / down_proj of a tensor-parallel decode step
// (minivllm/models/layers/linear.py:316-318 calls dist.all_reduce; RCCL's ring needs 2(W-1) latency-bound
// steps for a 1-2 MB message).  Two-shot, one launch, no host involvement, hipGraph-capturable:
//
//   every rank owns one UNCACHED device region, IPC-mapped by all peers:
//       staging[2][max_bytes] | flags[2 parities][2 phases][W ranks][NB blocks] | epoch, status
//   call number e (read from the region, so a replayed graph advances it), parity p = e & 1:
//     0. block b copies sub-chunk b of every slice of x into its own staging[p]
//     1. barrier(phase 0) with block b of every peer: "my sub-chunks b are in place"
//     2. block b of rank r sums sub-chunk b of slice r over all W ranks (fp32, rank order) and writes the
//        rounded result back into ITS OWN staging[p] (nobody else reads slice r of rank r's staging)
//     3. barrier(phase 1): "my reduced sub-chunk b is in place"
//     4. block b gathers sub-chunk b of slice j from rank j, for all j, into out
//   staging[p] is rewritten two calls later; a peer can only be two calls ahead after passing the
//   barriers of the call in between, i.e. after every rank finished reading — no trailing barrier.
//
// Cross-GPU visibility does not rely on cache state: the region is allocated uncached
// (hipDeviceMallocUncached), peer data is read with system-scope (sc0 sc1) loads, flags are
// system-scope atomics, and a system-scope fence separates data from flag on both sides.
// Every wait is BOUNDED in wall-clock time (s_memrealtime, 100 MHz): a peer that never arrives makes the kernel
// give up, count the expired wait in the status block and finish — the grid always drains.  The FIRST expired
// wait also leaves a record (which barrier phase, which peer, which block, the epoch it waited for and the flag
// value it last saw), so a time-out explains itself: a flag that still holds an OLD epoch means the peer never
// reached this call (not launched, dead, or descheduled for longer than the bound); a flag AHEAD of the epoch
// cannot expire.  The host reads the block after its validation run (falls back to RCCL) and, while serving,
// with every step's token copy (raises: a rank that continued with incomplete sums must not keep serving).
#include <string.h>

#include "common.h"

namespace {

constexpr int kMaxWorld = 8;
constexpr int kArBlocks = 32;       // workgroups per launch (= flag slots per rank and phase)
constexpr int kArThreads = 512;
constexpr uint32_t kDefaultWaitMs = 20000;   // bounded wait per barrier (wall clock); mv_ar_set_wait_limit_ms changes it
// status block (uint32 words at status_off): [0] expired waits so far  [1] first-record claimed
// [2] phase | peer << 8 | block << 16 | kernel << 24  [3] epoch waited for  [4] flag value last seen
// [5] wait limit in ms (host-written)
enum { kStExpired = 0, kStClaim = 1, kStWhere = 2, kStEpoch = 3, kStSeen = 4, kStLimitMs = 5, kStWords = 8 };

struct ArRegion {
  // byte offsets inside a rank's region
  size_t staging_bytes;  // per parity
  size_t flags_off, epoch_off, status_off, total;
};

__host__ __device__ inline ArRegion ar_layout(size_t max_bytes) {
  ArRegion r;
  r.staging_bytes = (max_bytes + 255) & ~(size_t)255;
  r.flags_off = 2 * r.staging_bytes;
  const size_t flags_bytes = (size_t)2 * 2 * kMaxWorld * kArBlocks * sizeof(uint32_t);
  r.epoch_off = r.flags_off + flags_bytes;
  r.status_off = r.epoch_off + 256;
  r.total = r.status_off + 256;
  return r;
}

struct ArParams {
  char *base[kMaxWorld];  // base[rank] = own region, base[q] = peer q's region mapped here
  const void *x;
  void *out;
  size_t max_bytes;
  int64_t n;  // elements
  int rank, world;
};

// system-scope (sc0 sc1) 16-byte load through a buffer descriptor: the compiler tracks its completion
// itself, so the W loads of one reduction step are all in flight before the first is consumed
typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t rsrc_t;
__device__ __forceinline__ auto make_rsrc(char *base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}
template <typename R>
__device__ __forceinline__ u32x4 load_sys(R rsrc, uint32_t byte_off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, /*sc0|sc1*/ 17));
}

__device__ __forceinline__ uint32_t *flag_ptr(char *base, const ArRegion &L, int parity, int phase, int src, int blk) {
  return (uint32_t *)(base + L.flags_off) + (((parity * 2 + phase) * kMaxWorld + src) * kArBlocks + blk);
}

// Block-level barrier with the same block of every peer.  Thread q < world handles peer q.
__device__ __forceinline__ void peer_barrier(const ArParams &p, const ArRegion &L, int parity, int phase,
                                             uint32_t epoch, uint32_t *status, int kernel_id = 0) {
  __threadfence_system();  // this thread's staging stores are performed before the flag goes out
  __syncthreads();
  const int q = threadIdx.x;
  if (q < p.world) {
    if (q != p.rank)
      __hip_atomic_store(flag_ptr(p.base[q], L, parity, phase, p.rank, blockIdx.x), epoch, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    if (q != p.rank) {
      uint32_t *mine = flag_ptr(p.base[p.rank], L, parity, phase, q, blockIdx.x);
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz, constant
      const uint64_t limit = (uint64_t)__hip_atomic_load(status + kStLimitMs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 100000ull;
      uint32_t seen;
      // epochs only grow; a flag from a later call (>= epoch + 2, same parity) also means "arrived"
      while ((int32_t)((seen = __hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) - epoch) < 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > limit) {
          __hip_atomic_fetch_add(status + kStExpired, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (__hip_atomic_exchange(status + kStClaim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
            status[kStWhere] = (uint32_t)phase | ((uint32_t)q << 8) | ((uint32_t)blockIdx.x << 16) | ((uint32_t)kernel_id << 24);
            status[kStEpoch] = epoch;
            status[kStSeen] = seen;
          }
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
  }
  __syncthreads();
  __threadfence_system();
}

template <typename Tag>
__global__ void __launch_bounds__(kArThreads) allreduce_two_shot_kernel(ArParams p) {
  const ArRegion L = ar_layout(p.max_bytes);
  char *self = p.base[p.rank];
  uint32_t *epoch_p = (uint32_t *)(self + L.epoch_off);
  uint32_t *status = (uint32_t *)(self + L.status_off);
  const uint32_t epoch = __hip_atomic_load(epoch_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const int parity = epoch & 1;
  const int W = p.world;
  // vectors of 8 elements; slice j = vectors [j*vs, (j+1)*vs), sub-chunk b = a contiguous vc-vector piece of it
  const int64_t nvec = p.n >> 3;
  const int64_t vs = nvec / W;             // host guarantees nvec % W == 0
  const int64_t vc = (vs + kArBlocks - 1) / kArBlocks;
  const int64_t c0 = (int64_t)blockIdx.x * vc, c1 = min(vs, c0 + vc);
  const size_t stage_off = (size_t)parity * L.staging_bytes;

  // 0. copy-in: sub-chunk b of every slice
  const u32x4 *xin = (const u32x4 *)p.x;
  u32x4 *mine = (u32x4 *)(self + stage_off);
  for (int j = 0; j < W; ++j)
    for (int64_t i = c0 + threadIdx.x; i < c1; i += kArThreads) mine[j * vs + i] = xin[j * vs + i];
  peer_barrier(p, L, parity, 0, epoch, status);

  // 2. reduce sub-chunk b of slice `rank` over all ranks, in rank order (all W loads in flight together)
  for (int64_t i = c0 + threadIdx.x; i < c1; i += kArThreads) {
    const uint32_t off = (uint32_t)(((int64_t)p.rank * vs + i) * 16);
    u32x4 v[kMaxWorld];
#pragma unroll
    for (int q = 0; q < kMaxWorld; ++q)
      if (q < W) v[q] = load_sys(make_rsrc(p.base[q] + stage_off, L.staging_bytes), off);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int q = 0; q < kMaxWorld; ++q)
      if (q < W) {
        float f[8];
        unpack8<Tag>(v[q], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f[e];
      }
    mine[(int64_t)p.rank * vs + i] = pack8<Tag>(acc);
  }
  peer_barrier(p, L, parity, 1, epoch, status);

  // 4. gather the reduced sub-chunk b of slice j from rank j
  u32x4 *out = (u32x4 *)p.out;
  for (int64_t i = c0 + threadIdx.x; i < c1; i += kArThreads) {
    u32x4 v[kMaxWorld];
#pragma unroll
    for (int j = 0; j < kMaxWorld; ++j)
      if (j < W) v[j] = load_sys(make_rsrc(p.base[j] + stage_off, L.staging_bytes), (uint32_t)(((int64_t)j * vs + i) * 16));
#pragma unroll
    for (int j = 0; j < kMaxWorld; ++j)
      if (j < W) out[j * vs + i] = v[j];
  }

  // advance the call counter once every block of this rank is past its last flag use
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t *done = epoch_p + 1;
    const uint32_t t = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (uint32_t)gridDim.x - 1) {
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(epoch_p, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// f1 — all-reduce + residual add + RMSNorm in ONE launch, for a [M, K] activation with
// M % world == 0: the same two-shot protocol with ROWS as the unit.  Row r belongs to slice r / (M/W) (the
// rank that reduces it) and to block r % NB on every rank (so the per-block barriers cover every hand-off);
// inside a block, wave (r / NB) % 8 owns it: a whole row per wave, the row norm is a wave reduction, no LDS.
//   0. copy my rows into staging   1. barrier   2. reduce the rows of my slice (fp32, rank order, -> dtype)
//   3. barrier   4. per row: fetch the reduced row from its owner, s = f32(sum) + f32(residual),
//      residual <- s -> dtype (in place), out <- ((s * rsqrt(mean(s^2)+eps)) * w) -> dtype
// i.e. dist.all_reduce followed by RMSNorm.add_rms_forward (minivllm/models/layers/linear.py:316-318,
// layers/layernorm.py:95-105) with the rounding points of the unfused pair (the sum is rounded to dtype
// before the residual is added).  norm_w == nullptr: plain row-wise all-reduce into out.
constexpr int kRowVecMax = 16; // 16-byte vectors per lane and row: K <= 8192 (template NV = 4, 8 or 16; at NV = 16 —
                               // Qwen2-72B's hidden size — a wave holds 2 x 64 packed registers of one row, inside
                               // the 256 a 512-thread workgroup's waves may use)

struct ArRowParams {
  ArParams ar;        // x = [M][K] partial sums (row stride K), out = normed output (or the plain sum)
  void *residual;     // [M][K], in/out; nullptr with norm_w == nullptr
  const void *norm_w; // [K]
  int M, K;
  float eps;
};

template <typename Tag, int NV>
__global__ void __launch_bounds__(kArThreads) allreduce_rows_kernel(ArRowParams rp) {
  const ArParams &p = rp.ar;
  const ArRegion L = ar_layout(p.max_bytes);
  char *self = p.base[p.rank];
  uint32_t *epoch_p = (uint32_t *)(self + L.epoch_off);
  uint32_t *status = (uint32_t *)(self + L.status_off);
  const uint32_t epoch = __hip_atomic_load(epoch_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const int parity = epoch & 1;
  const int W = p.world, NB = gridDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = kArThreads / 64;
  const int nv = rp.K >> 3;                 // vectors per row
  const int Ms = rp.M / W;                  // rows per slice
  const size_t stage_off = (size_t)parity * L.staging_bytes;
  u32x4 *mine = (u32x4 *)(self + stage_off);
  const u32x4 *xin = (const u32x4 *)p.x;

  // rows of this (block, wave): r = blockIdx.x + NB * (wave + nw * i)
  // 0. copy-in
  for (int r = blockIdx.x + NB * wave; r < rp.M; r += NB * nw)
    for (int v = lane; v < nv; v += 64) mine[(int64_t)r * nv + v] = xin[(int64_t)r * nv + v];
  peer_barrier(p, L, parity, 0, epoch, status, 1);

  // 2. reduce the rows of slice `rank`
  for (int r = blockIdx.x + NB * wave; r < rp.M; r += NB * nw) {
    if (r / Ms != p.rank) continue;
    for (int v = lane; v < nv; v += 64) {
      const uint32_t off = (uint32_t)(((int64_t)r * nv + v) * 16);
      u32x4 in[kMaxWorld];
#pragma unroll
      for (int q = 0; q < kMaxWorld; ++q)
        if (q < W) in[q] = load_sys(make_rsrc(p.base[q] + stage_off, L.staging_bytes), off);
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
      for (int q = 0; q < kMaxWorld; ++q)
        if (q < W) {
          float f[8];
          unpack8<Tag>(in[q], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
      mine[(int64_t)r * nv + v] = pack8<Tag>(acc);
    }
  }
  peer_barrier(p, L, parity, 1, epoch, status, 1);

  // 4. gather (+ residual add + RMSNorm): one row per wave at a time, the row stays in registers
  u32x4 *out = (u32x4 *)p.out;
  u32x4 *res = (u32x4 *)rp.residual;
  const u32x4 *nwp = (const u32x4 *)rp.norm_w;
  for (int r = blockIdx.x + NB * wave; r < rp.M; r += NB * nw) {
    const int owner = r / Ms;
    u32x4 a[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = lane + 64 * i;
      if (v < nv) {
        a[i] = load_sys(make_rsrc(p.base[owner] + stage_off, L.staging_bytes), (uint32_t)(((int64_t)r * nv + v) * 16));
        if (NV <= 8 && nwp) b[i] = res[(int64_t)r * nv + v];
      }
    }
    if (!nwp) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < nv) out[(int64_t)r * nv + lane + 64 * i] = a[i];
      continue;
    }
    float ss = 0.f;
    if constexpr (NV > 8) {
      // wide rows (K = 8192): only the reduced row stays in registers (64 packed); the residual is read twice
      // from this GPU's own L2 instead — once for the row norm, once to form the sum that is stored and scaled
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < nv) {
          float fa[8], fb[8];
          unpack8<Tag>(a[i], fa);
          unpack8<Tag>(res[(int64_t)r * nv + lane + 64 * i], fb);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            fa[e] = __fadd_rn(fa[e], fb[e]);
            ss = fmaf(fa[e], fa[e], ss);
          }
        }
      ss = wave_sum(ss);
      const float rs = 1.0f / sqrtf(ss / (float)rp.K + rp.eps);
      asm volatile("" ::: "memory");   // the second residual read and the weight read start here, not above
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < nv) {
          float fa[8], fb[8], fw[8], y[8];
          unpack8<Tag>(a[i], fa);
          unpack8<Tag>(res[(int64_t)r * nv + lane + 64 * i], fb);
          unpack8<Tag>(nwp[lane + 64 * i], fw);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            fa[e] = __fadd_rn(fa[e], fb[e]);
            y[e] = __fmul_rn(__fmul_rn(fa[e], rs), fw[e]);
          }
          res[(int64_t)r * nv + lane + 64 * i] = pack8<Tag>(fa);  // residual <- (sum + residual) -> dtype
          out[(int64_t)r * nv + lane + 64 * i] = pack8<Tag>(y);
        }
      continue;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nv) {
        float fa[8], fb[8];
        unpack8<Tag>(a[i], fa);
        unpack8<Tag>(b[i], fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          fa[e] = __fadd_rn(fa[e], fb[e]);
          ss = fmaf(fa[e], fa[e], ss);
        }
        res[(int64_t)r * nv + lane + 64 * i] = pack8<Tag>(fa);  // residual <- (sum + residual) -> dtype
      }
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)rp.K + rp.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nv) {
        float fa[8], fb[8], fw[8], y[8];
        unpack8<Tag>(a[i], fa);
        unpack8<Tag>(b[i], fb);
        unpack8<Tag>(nwp[lane + 64 * i], fw);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = __fmul_rn(__fmul_rn(__fadd_rn(fa[e], fb[e]), rs), fw[e]);
        out[(int64_t)r * nv + lane + 64 * i] = pack8<Tag>(y);
      }
  }

  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t *done = epoch_p + 1;
    const uint32_t t = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (uint32_t)gridDim.x - 1) {
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(epoch_p, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// f2: greedy decoding under tensor parallelism without gathering the logits.  Every rank packs,
// per row, the maximum of its vocab shard and the global index of that maximum into four 16-bit floats that
// are EXACT in bf16/f16 (the logit itself; three 8-bit pieces of the index) at slot [rank] of a zeroed
// [world][rows][4] buffer; the all-reduce above (sum with zeros = gather, exact) then hands every rank all
// candidates, and the merge keeps the largest value, lowest global index on ties — argmax of the full row.
template <typename Tag>
__global__ void greedy_pack_kernel(const uint16_t *logits, int64_t ld, const int64_t *local_tokens, int rank, int world,
                                   int64_t vocab_per_rank, int rows, int rows_padded, uint16_t *buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over world * rows_padded
  if (i >= world * rows_padded) return;
  const int q = i / rows_padded, r = i - q * rows_padded;
  u32x2 w = {0u, 0u};
  if (q == rank && r < rows) {
    const int64_t t = local_tokens[r], g = t + (int64_t)rank * vocab_per_rank;
    const uint32_t val = logits[(int64_t)r * ld + t];
    w[0] = val | ((uint32_t)elem<Tag>::st((float)((g >> 16) & 255)) << 16);
    w[1] = (uint32_t)elem<Tag>::st((float)((g >> 8) & 255)) | ((uint32_t)elem<Tag>::st((float)(g & 255)) << 16);
  }
  *(u32x2 *)(buf + (int64_t)i * 4) = w;
}

template <typename Tag>
__global__ void greedy_merge_kernel(const uint16_t *buf, int world, int rows, int rows_padded, int64_t *tokens) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float best = -INFINITY;
  int64_t best_idx = INT64_MAX;
  for (int q = 0; q < world; ++q) {
    const u32x2 w = *(const u32x2 *)(buf + ((int64_t)q * rows_padded + r) * 4);
    const float v = elem<Tag>::ld((uint16_t)(w[0] & 0xffffu));
    const int64_t g = ((int64_t)elem<Tag>::ld((uint16_t)(w[0] >> 16)) << 16) |
                      ((int64_t)elem<Tag>::ld((uint16_t)(w[1] & 0xffffu)) << 8) |
                      (int64_t)elem<Tag>::ld((uint16_t)(w[1] >> 16));
    if (v > best || (v == best && g < best_idx) || best_idx == INT64_MAX) {
      best = v;
      best_idx = g;
    }
  }
  tokens[r] = best_idx;
}

}  // namespace

extern "C" int mv_ar_all_reduce_add_rmsnorm(const void *x, void *residual, const void *norm_w, float eps, void *out,
                                            int m, int k, void *const *bases, int rank, int world,
                                            size_t max_bytes, int dtype, void *stream) {
  if (m == 0) return MV_OK;
  if (!x || !out || !bases || (norm_w && !residual)) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (world < 2 || world > kMaxWorld || rank < 0 || rank >= world) return MV_E_RANGE;
  if (m < 0 || k <= 0 || (m % world) != 0 || (k & 7) || k > 8 * 64 * kRowVecMax ||
      (size_t)m * (size_t)k * 2 > max_bytes)
    return MV_E_SHAPE;
  if (!mv_aligned16(x) || !mv_aligned16(out) || (residual && !mv_aligned16(residual)) ||
      (norm_w && !mv_aligned16(norm_w)))
    return MV_E_ALIGN;
  ArRowParams rp;
  for (int q = 0; q < kMaxWorld; ++q) rp.ar.base[q] = q < world ? (char *)bases[q] : nullptr;
  for (int q = 0; q < world; ++q)
    if (!rp.ar.base[q]) return MV_E_NULLPTR;
  rp.ar.x = x;
  rp.ar.out = out;
  rp.ar.max_bytes = max_bytes;
  rp.ar.n = (int64_t)m * k;
  rp.ar.rank = rank;
  rp.ar.world = world;
  rp.residual = residual;
  rp.norm_w = norm_w;
  rp.M = m;
  rp.K = k;
  rp.eps = eps;
  const int nvl = (k / 8 + 63) / 64;  // vectors per lane and row
  const dim3 grid(kArBlocks), block(kArThreads);
  hipStream_t st = (hipStream_t)stream;
#define MV_AR_ROWS(TAG)                                                                                   \
  do {                                                                                                    \
    if (nvl <= 4) hipLaunchKernelGGL((allreduce_rows_kernel<TAG, 4>), grid, block, 0, st, rp);          \
    else if (nvl <= 8) hipLaunchKernelGGL((allreduce_rows_kernel<TAG, 8>), grid, block, 0, st, rp);     \
    else hipLaunchKernelGGL((allreduce_rows_kernel<TAG, 16>), grid, block, 0, st, rp);                  \
  } while (0)
  if (dtype == MV_BF16) MV_AR_ROWS(bf16_tag);
  else MV_AR_ROWS(f16_tag);
#undef MV_AR_ROWS
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_greedy_pack(const void *logits_local, int64_t ld, const int64_t *local_tokens, int rank, int world,
                              int64_t vocab_per_rank, int rows, int rows_padded, void *buf, int dtype,
                              void *stream) {
  if (rows == 0) return MV_OK;
  if (!logits_local || !local_tokens || !buf) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (rows < 0 || rows_padded < rows || world < 1 || rank < 0 || rank >= world ||
      vocab_per_rank * world > (1 << 24))
    return MV_E_SHAPE;
  if (((uintptr_t)buf) & 7u) return MV_E_ALIGN;
  const int n = world * rows_padded;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((greedy_pack_kernel<bf16_tag>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)logits_local, ld, local_tokens, rank, world, vocab_per_rank, rows,
                       rows_padded, (uint16_t *)buf);
  else
    hipLaunchKernelGGL((greedy_pack_kernel<f16_tag>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)logits_local, ld, local_tokens, rank, world, vocab_per_rank, rows,
                       rows_padded, (uint16_t *)buf);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" int mv_greedy_merge(const void *buf, int world, int rows, int rows_padded, int64_t *tokens, int dtype,
                               void *stream) {
  if (rows == 0) return MV_OK;
  if (!buf || !tokens) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (rows < 0 || rows_padded < rows || world < 1) return MV_E_SHAPE;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((greedy_merge_kernel<bf16_tag>), dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)buf, world, rows, rows_padded, tokens);
  else
    hipLaunchKernelGGL((greedy_merge_kernel<f16_tag>), dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)buf, world, rows, rows_padded, tokens);
  MV_LAUNCH_CHECK();
  return MV_OK;
}

extern "C" size_t mv_ar_region_bytes(size_t max_bytes) { return ar_layout(max_bytes).total; }
extern "C" int mv_ar_max_world(void) { return kMaxWorld; }

extern "C" int mv_ar_alloc(size_t max_bytes, void **base_out) {
  if (!base_out || max_bytes == 0) return MV_E_NULLPTR;
  const size_t total = ar_layout(max_bytes).total;
  void *p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, total, hipDeviceMallocUncached);
  if (e != hipSuccess) return (int)e;
  e = hipMemset(p, 0, total);
  if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
  const uint32_t limit = kDefaultWaitMs;
  e = hipMemcpy((char *)p + ar_layout(max_bytes).status_off + kStLimitMs * sizeof(uint32_t), &limit, sizeof(limit),
                hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
  e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
  *base_out = p;
  return MV_OK;
}

// how long a barrier of this rank's launches waits for a peer before giving up (default 20 s)
extern "C" int mv_ar_set_wait_limit_ms(void *base, size_t max_bytes, uint32_t ms) {
  if (!base) return MV_E_NULLPTR;
  if (ms == 0) return MV_E_RANGE;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return (int)e;
  return mv_check_hip(hipMemcpy((char *)base + ar_layout(max_bytes).status_off + kStLimitMs * sizeof(uint32_t), &ms,
                                sizeof(ms), hipMemcpyHostToDevice));
}

extern "C" int mv_ar_free(void *base) { return base ? mv_check_hip(hipFree(base)) : MV_OK; }

extern "C" int mv_ar_ipc_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

extern "C" int mv_ar_ipc_export(void *base, void *handle_out) {
  if (!base || !handle_out) return MV_E_NULLPTR;
  return mv_check_hip(hipIpcGetMemHandle((hipIpcMemHandle_t *)handle_out, base));
}

extern "C" int mv_ar_ipc_open(const void *handle, void **ptr_out) {
  if (!handle || !ptr_out) return MV_E_NULLPTR;
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  return mv_check_hip(hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess));
}

extern "C" int mv_ar_ipc_close(void *ptr) { return ptr ? mv_check_hip(hipIpcCloseMemHandle(ptr)) : MV_OK; }

// status block of the own region (8 words, see kSt*): word 0 = 0 while every wait so far was satisfied.
// Host call, synchronises the device.
extern "C" int mv_ar_status(void *base, size_t max_bytes, uint32_t *status_out) {
  if (!base || !status_out) return MV_E_NULLPTR;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return (int)e;
  return mv_check_hip(hipMemcpy(status_out, (char *)base + ar_layout(max_bytes).status_off, kStWords * sizeof(uint32_t),
                                hipMemcpyDeviceToHost));
}

// the same 8 words copied asynchronously on `stream` into pinned host memory: rides along with a decode step's
// token copy, so the host sees a time-out with the step it happened in, without an extra synchronisation
extern "C" int mv_ar_status_async(void *base, size_t max_bytes, uint32_t *pinned_host_out, void *stream) {
  if (!base || !pinned_host_out) return MV_E_NULLPTR;
  return mv_check_hip(hipMemcpyAsync(pinned_host_out, (char *)base + ar_layout(max_bytes).status_off,
                                     kStWords * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
}

extern "C" int mv_ar_all_reduce(const void *x, void *out, int64_t n, void *const *bases, int rank, int world,
                                size_t max_bytes, int dtype, void *stream) {
  if (n == 0) return MV_OK;
  if (!x || !out || !bases) return MV_E_NULLPTR;
  if (dtype != MV_BF16 && dtype != MV_F16) return dtype == MV_F32 ? MV_E_SHAPE : MV_E_DTYPE;
  if (world < 2 || world > kMaxWorld || rank < 0 || rank >= world) return MV_E_RANGE;
  if (n < 0 || (n % (8 * world)) != 0 || (size_t)n * 2 > max_bytes) return MV_E_SHAPE;
  if (!mv_aligned16(x) || !mv_aligned16(out)) return MV_E_ALIGN;
  ArParams p;
  for (int q = 0; q < kMaxWorld; ++q) p.base[q] = q < world ? (char *)bases[q] : nullptr;
  for (int q = 0; q < world; ++q)
    if (!p.base[q]) return MV_E_NULLPTR;
  p.x = x;
  p.out = out;
  p.max_bytes = max_bytes;
  p.n = n;
  p.rank = rank;
  p.world = world;
  if (dtype == MV_BF16)
    hipLaunchKernelGGL((allreduce_two_shot_kernel<bf16_tag>), dim3(kArBlocks), dim3(kArThreads), 0,
                       (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((allreduce_two_shot_kernel<f16_tag>), dim3(kArBlocks), dim3(kArThreads), 0,
                       (hipStream_t)stream, p);
  MV_LAUNCH_CHECK();
  return MV_OK;
}
