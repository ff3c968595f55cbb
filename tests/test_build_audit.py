"""Build audit of the flash prefill kernel (host test; hipcc cross-compiles gfx950 without a GPU).

The default (PIPE) ``prefill_mfma_kernel`` retires its inline-assembly LDS reads with HAND-COUNTED ``s_waitcnt
lgkmcnt(N)`` (csrc/prefill_mfma.hip).  ``lgkmcnt`` also counts scalar memory loads, so a count is only sound while the
compiler places no ``s_load`` / ``s_buffer_load`` inside the span of the counted reads; and the kernel must neither
spill nor use scratch.  A compiler or code change could silently break either: this test re-checks the
four shipped instantiations and the 64-rows-per-wave bench kernel (its accumulator file is asm-owned) on every run."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = tmp_path_factory.mktemp("audit") / "prefill_mfma.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-fno-slp-vectorize",
                    "--cuda-device-only", "-S", "-o", str(out), os.path.join(ROOT, "mini-vllm_amd", "csrc", "prefill_mfma.hip")],
                   check=True, stderr=subprocess.DEVNULL)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_ZN\S*prefill_mfma\S*kernel\S*):\s*; @", text, re.M):
        end = text.index("; ScratchSize", m.end())
        kernels[m.group(1)] = text[m.end():text.index("\n", end)]
    return kernels


def _facts(body):
    return {k: int(re.search(rf"; {k}: (\d+)", body).group(1)) for k in ("NumVgprs", "ScratchSize")}


def test_pipe_prefill_kernels_keep_scalar_loads_out_of_the_counted_lds_reads(asm):
    pipe = {n: b for n, b in asm.items() if "prefill_mfma_kernelI" in n and "Lb1E" in n}
    assert len(pipe) == 4, sorted(asm)                       # bf16 / f16 x 4 / 8 waves
    for name, body in pipe.items():
        lines = body.split("\n")
        # the counted waits are the hand-written ones: inside ;;#ASMSTART ... ;;#ASMEND blocks
        in_asm, counted = False, []
        for i, ln in enumerate(lines):
            if ";;#ASMSTART" in ln:
                in_asm = True
            elif ";;#ASMEND" in ln:
                in_asm = False
            elif in_asm and re.search(r"s_waitcnt\s+lgkmcnt\([1-9]\d*\)", ln):
                counted.append(i)
        reads = [i for i, ln in enumerate(lines) if re.search(r"\bds_read", ln)]
        assert counted and reads, name
        lo, hi = min(reads), max(max(reads), max(counted))
        scalar = [lines[i].strip() for i in range(lo, hi + 1) if re.search(r"\bs_(buffer_)?load_", lines[i])]
        assert not scalar, f"{name}: scalar loads inside the counted span: {scalar[:4]}"
        f = _facts(body)
        assert f["ScratchSize"] == 0 and f["NumVgprs"] <= 256, (name, f)


def test_64_row_prefill_kernel_owns_its_accumulator_file(asm):
    """bench variant 5 (one wave per SIMD, a[0:191] named in inline asm): no compiler accumulator traffic outside the asm
    blocks, no scratch (what tools/dbg/audit_pf64.sh checked by hand in round 3)."""
    k64 = {n: b for n, b in asm.items() if "prefill_mfma64_kernelI" in n and "Li0E" in n}
    assert k64
    for name, body in k64.items():
        out_of_asm, in_asm = [], False
        for ln in body.split("\n"):
            if ";;#ASMSTART" in ln:
                in_asm = True
            elif ";;#ASMEND" in ln:
                in_asm = False
            elif not in_asm:
                out_of_asm.append(ln)
        touched = [ln.strip() for ln in out_of_asm if re.search(r"accvgpr| a\[\d|, a\d", ln)]
        assert not touched, (name, touched[:4])
        assert _facts(body)["ScratchSize"] == 0 and "scratch_" not in body, name
