"""Small-message all-reduce over xGMI peer-to-peer loads (csrc/allreduce.hip) for the tensor-parallel
decode step: the [tokens, hidden] sums after o_proj / down_proj and the embedding
(minivllm/models/layers/linear.py:316-318, layers/embed_head.py:170-173 call ``dist.all_reduce``).

Set-up is collective and conservative: every rank allocates one uncached region, the IPC handles are
exchanged through the process group, and the kernel is then VALIDATED against the backend's own
all-reduce on exact (integer-valued) data, several calls back to back.  Only if every rank reports
success is the kernel used; any failure (allocation, IPC, a wait that gave up, a mismatch) leaves RCCL
in charge.  ``MV_CUSTOM_AR=0`` disables it.

While serving, a peer wait that gives up (a stalled or dead rank) must not go unnoticed: the kernel then
finishes with incomplete sums, and ranks that sample different tokens fall out of the lock-step SPMD schedule.
The status block of the region therefore travels to the host with every step's token copy
(``post_status`` / ``verify``); a non-zero count raises ``PeerWaitExpired`` with the record of the first
expired wait (phase, peer, block, epoch waited for, flag value seen)."""
from __future__ import annotations

import ctypes
import logging
import os

import torch
import torch.distributed as dist

from . import hip_ops

logger = logging.getLogger(__name__)

__all__ = ["CustomAllReduce", "PeerWaitExpired", "maybe_create", "describe_status"]


class PeerWaitExpired(RuntimeError):
    """A bounded wait of the peer-to-peer all-reduce gave up: this rank's sums are incomplete."""


def describe_status(words) -> str:
    """Human-readable form of the 8-word status block (csrc/allreduce.hip: kSt*)."""
    w = [int(x) & 0xFFFFFFFF for x in words]
    if w[0] == 0:
        return "ok"
    where = w[2]
    kernel = {0: "all_reduce", 1: "all_reduce_add_rmsnorm"}.get(where >> 24, str(where >> 24))
    seen, want = w[4], w[3]
    lag = (want - seen) & 0xFFFFFFFF
    return (f"{w[0]} peer wait(s) expired after {w[5]} ms; first: kernel {kernel}, barrier phase {where & 255} "
            f"('{'inputs staged' if (where & 255) == 0 else 'slice reduced'}'), waiting for rank {(where >> 8) & 255}, "
            f"block {(where >> 16) & 255}, call #{want}; the peer's flag still held #{seen} "
            f"({lag} call(s) behind: the peer never reached this call)")

MAX_BYTES = 8 << 20           # largest message taken (256 tokens x 8192 x 2 B = 4 MiB for Qwen2-72B)
VALIDATION_WAIT_MS = 3000     # bound of a peer wait during the start-up validation (ranks sharing ONE card are time-sliced)
SERVING_WAIT_MS = 20000       # csrc/allreduce.hip: kDefaultWaitMs
_lib = hip_ops._lib


def _all_ok(ok: bool, group) -> bool:
    """Collective AND over the ranks (host side)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        flag = flag.cuda()
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item())


class CustomAllReduce:
    def __init__(self, rank: int, world: int, device: torch.device, group=None, max_bytes: int = MAX_BYTES) -> None:
        self.rank, self.world, self.device, self.group = rank, world, device, group
        self.max_bytes = int(max_bytes)
        self.base = ctypes.c_void_p()
        self.peers: list[int | None] = [None] * world
        self.bases = (ctypes.c_void_p * 8)()
        self.ready = False
        self.calls = 0
        self._status_ring = None   # pinned [slots, 8] uint32 (as int64-safe int32 bits) for post_status / verify
        self._status_slot = 0

    # ------------------------------------------------------------------ collective set-up
    def setup(self) -> bool:
        g = self.group
        ok = True
        handle = b""
        try:
            if self.world > int(_lib.mv_ar_max_world()):
                raise RuntimeError(f"world size {self.world} not supported")
            hip_ops._check(_lib.mv_ar_alloc(self.max_bytes, ctypes.byref(self.base)), "mv_ar_alloc")
            buf = ctypes.create_string_buffer(int(_lib.mv_ar_ipc_handle_bytes()))
            hip_ops._check(_lib.mv_ar_ipc_export(self.base, buf), "mv_ar_ipc_export")
            handle = buf.raw
        except Exception as e:  # noqa: BLE001 - any local failure only disables the fast path
            logger.warning("custom all-reduce: allocation/export failed on rank %d: %s", self.rank, e)
            ok = False
        if not _all_ok(ok, g):
            return self._give_up()
        handles: list = [None] * self.world
        dist.all_gather_object(handles, handle, group=g)
        try:
            for q, h in enumerate(handles):
                if q == self.rank:
                    self.bases[q] = self.base.value
                    continue
                ptr = ctypes.c_void_p()
                hip_ops._check(_lib.mv_ar_ipc_open(ctypes.create_string_buffer(h, len(h)), ctypes.byref(ptr)),
                               "mv_ar_ipc_open")
                self.peers[q] = ptr.value
                self.bases[q] = ptr.value
            # load this library's code object NOW (HIP loads it at the first launch): a rank that spends
            # seconds on that inside the first collective call would make its peers' bounded waits give up
            hip_ops.greedy_merge(torch.zeros((self.world, 2, 4), dtype=torch.bfloat16, device=self.device), 1)
            torch.cuda.synchronize(self.device)
        except Exception as e:  # noqa: BLE001
            logger.warning("custom all-reduce: IPC open failed on rank %d: %s", self.rank, e)
            ok = False
        if not _all_ok(ok, g):
            return self._give_up()
        try:
            # a peer wait that expires during the validation means "do not use this kernel", not "wait again": a short
            # bound here (the code object is loaded, every rank is inside the same call), the serving bound afterwards —
            # 65 validation calls at the serving bound would be twenty minutes of start-up on a box whose peer loads
            # cannot see each other
            self.set_wait_limit_ms(VALIDATION_WAIT_MS)
            ok = self._validate()
            self.set_wait_limit_ms(SERVING_WAIT_MS)
        except Exception as e:  # noqa: BLE001
            logger.warning("custom all-reduce: validation raised on rank %d: %s", self.rank, e)
            ok = False
        if not _all_ok(ok, g):
            return self._give_up()
        self.ready = True
        logger.info("custom all-reduce enabled (rank %d of %d, messages up to %d KiB)", self.rank, self.world,
                    self.max_bytes >> 10)
        return True

    def _give_up(self) -> bool:
        logger.warning("custom all-reduce disabled on rank %d: the backend's all-reduce stays in charge", self.rank)
        self.close()
        return False

    def _validate(self) -> bool:
        """Exact comparison with the backend's all-reduce: integer-valued data sums exactly in bf16/f16 in any
        order.  Several sizes, several calls in a row without a host sync in between (flag parity reuse)."""
        gen = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        good = True
        for dtype in (torch.bfloat16, torch.float16):
            for n in (8 * self.world, 1024 * self.world, 256 * 2048, 256 * 3584, self.max_bytes // 2):
                n -= n % (8 * self.world)
                pairs = []
                for _ in range(5):
                    x = torch.randint(-6, 7, (n,), generator=gen).to(dtype).to(self.device)
                    want = x.clone()
                    got = self._launch(x, torch.empty_like(x))
                    inplace = self._launch(x.clone(), None)
                    pairs.append((want, got, inplace))
                for i, (want, got, inplace) in enumerate(pairs):
                    dist.all_reduce(want, group=self.group)
                    ok = torch.equal(got, want) and torch.equal(inplace, want)
                    if not ok and good:
                        logger.warning("custom all-reduce validation: mismatch on rank %d (%s, n=%d, call %d, "
                                       "%d / %d elements differ; %s)", self.rank, dtype, n, i,
                                       int((got != want).sum()), int((inplace != want).sum()),
                                       describe_status(self.status_words()))
                    good = good and ok
                # a wait that gave up anywhere ends the validation HERE, on every rank together (the decision is itself
                # a collective, so the ranks' backend calls stay aligned): the remaining sizes would only time out again
                mine = self.status_words()
                if mine[0] != 0:
                    logger.warning("custom all-reduce validation on rank %d: %s", self.rank, describe_status(mine))
                if not _all_ok(mine[0] == 0, self.group):
                    return False
        # the fused all-reduce + add + RMSNorm launch against the unfused pair (exact sums -> identical inputs
        # to the norm; the row norm itself may differ in the last bit of rstd: compare within 2 ulp)
        for dtype, (m, k) in ((torch.bfloat16, (8 * self.world, 2048)), (torch.float16, (self.world, 3584)),
                              (torch.bfloat16, (256, 2048)), (torch.bfloat16, (256, 8192)), (torch.float16, (64, 8192))):
            if m % self.world:
                m += self.world - m % self.world
            for _ in range(3):
                x = torch.randint(-6, 7, (m, k), generator=gen).to(dtype).to(self.device)
                res = torch.randn((m, k), generator=gen).to(dtype).to(self.device)
                w = (1 + 0.1 * torch.randn(k, generator=gen)).to(dtype).to(self.device)
                summed = x.clone()
                dist.all_reduce(summed, group=self.group)
                want, want_res = hip_ops.add_rmsnorm(summed, res.clone(), w, 1e-6)
                got, got_res = self.all_reduce_add_rmsnorm(x, res.clone(), w, 1e-6)
                eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
                ok = (torch.equal(got_res, want_res)
                      and bool(((got.float() - want.float()).abs() <= 2 * eps * want.float().abs() + 1e-6).all()))
                if not ok and good:
                    logger.warning("custom all-reduce validation: fused add+RMSNorm mismatch on rank %d (%s, %dx%d, "
                                   "residual equal: %s; %s)", self.rank, dtype, m, k,
                                   torch.equal(got_res, want_res), describe_status(self.status_words()))
                good = good and ok
        words = self.status_words()
        if words[0] != 0:
            logger.warning("custom all-reduce validation on rank %d: %s", self.rank, describe_status(words))
        return good and words[0] == 0

    # ------------------------------------------------------------------ use
    def accepts(self, x: torch.Tensor) -> bool:
        return (self.ready and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.is_contiguous()
                and 0 < x.numel() * 2 <= self.max_bytes and x.numel() % (8 * self.world) == 0
                and x.data_ptr() % 16 == 0)

    def _launch(self, x: torch.Tensor, out: torch.Tensor | None) -> torch.Tensor:
        out = x if out is None else out
        hip_ops._check(_lib.mv_ar_all_reduce(x.data_ptr(), out.data_ptr(), x.numel(), self.bases, self.rank,
                                             self.world, self.max_bytes, hip_ops._dt(x), hip_ops._stream()),
                       "mv_ar_all_reduce")
        self.calls += 1
        return out

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        """In-place SUM over the ranks (``accepts(x)`` must hold)."""
        return self._launch(x, None)

    def accepts_rows(self, x: torch.Tensor) -> bool:
        """[M, K] activations the fused all-reduce + add + RMSNorm launch takes."""
        return (self.accepts(x) and x.dim() == 2 and x.shape[0] % self.world == 0 and x.shape[1] % 8 == 0
                and x.shape[1] <= 8192)

    def all_reduce_add_rmsnorm(self, x, residual, norm_weight, eps: float):
        """-> (normed, residual): ``x`` summed over the ranks, added to ``residual`` (updated in place),
        RMS-normalised (``accepts_rows(x)`` must hold) -- dist.all_reduce + RMSNorm.add_rms_forward, one launch."""
        out = torch.empty_like(x)
        nw = norm_weight if norm_weight.dtype == x.dtype else norm_weight.to(x.dtype)
        hip_ops._check(_lib.mv_ar_all_reduce_add_rmsnorm(
            x.data_ptr(), residual.data_ptr(), nw.data_ptr(), float(eps), out.data_ptr(), x.shape[0], x.shape[1],
            self.bases, self.rank, self.world, self.max_bytes, hip_ops._dt(x), hip_ops._stream()),
            "mv_ar_all_reduce_add_rmsnorm")
        self.calls += 1
        return out, residual

    def status_words(self) -> list[int]:
        """The 8-word status block (host call: synchronises the device); word 0 = expired waits so far."""
        st = (ctypes.c_uint32 * 8)()
        hip_ops._check(_lib.mv_ar_status(self.base, self.max_bytes, st), "mv_ar_status")
        return [int(v) for v in st]

    def status(self) -> int:
        """0 while every peer wait of every call so far was satisfied (host call: synchronises the device)."""
        return self.status_words()[0]

    def set_wait_limit_ms(self, ms: int) -> None:
        hip_ops._check(_lib.mv_ar_set_wait_limit_ms(self.base, self.max_bytes, int(ms)), "mv_ar_set_wait_limit_ms")

    # -- per-step check without an extra synchronisation --------------------------------------------------
    def post_status(self) -> int:
        """Enqueue an async copy of the status block on the current stream into a pinned ring slot; returns the
        slot.  Call right after the step's work was enqueued (next to the token copy)."""
        if self._status_ring is None:
            self._status_ring = torch.zeros((8, 8), dtype=torch.int32, pin_memory=True)
        slot = self._status_slot
        self._status_slot = (slot + 1) % self._status_ring.shape[0]
        hip_ops._check(_lib.mv_ar_status_async(self.base, self.max_bytes, self._status_ring[slot].data_ptr(),
                                               hip_ops._stream()), "mv_ar_status_async")
        return slot

    def verify(self, slot: int) -> None:
        """After the stream work up to ``post_status`` is known complete (the step's tokens were read): raise if
        any peer wait has expired — this rank's activations are then sums over a subset of the ranks."""
        words = self._status_ring[slot].tolist()
        if words[0] != 0:
            self.ready = False
            raise PeerWaitExpired(f"tensor-parallel all-reduce on rank {self.rank}: {describe_status(words)}; "
                                  "the results of this step are invalid, stopping")

    def close(self) -> None:
        self.ready = False
        for q, ptr in enumerate(self.peers):
            if ptr:
                _lib.mv_ar_ipc_close(ctypes.c_void_p(ptr))
                self.peers[q] = None
        if self.base:
            _lib.mv_ar_free(self.base)
            self.base = ctypes.c_void_p()


def maybe_create(rank: int, world: int, device: torch.device, group=None) -> CustomAllReduce | None:
    """Collective: every rank of ``group`` must call it.  Returns the ready object, or None."""
    if world < 2 or os.environ.get("MV_CUSTOM_AR", "1") == "0" or device.type != "cuda":
        return None
    ar = CustomAllReduce(rank, world, device, group)
    return ar if ar.setup() else None
