"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/bcbf.h declares, and the host wrappers refuse to run without a GPU (no fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from bayesian_cbf_amd.build import build
    build()
    from bayesian_cbf_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "bcbf.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(bcbf_\w+)\s*\(", header))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib.lib, name), "libbcbf.so does not export %s" % name
    assert declared == set(lib.declared_symbols())
    # ... and nothing else: the library is built with hidden visibility, internal cross-file entry points stay internal
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert exported == declared, sorted(exported ^ declared)


def test_version_and_layout_helpers(lib):
    assert lib.lib.bcbf_version() == 1
    # Np*(Np+2)/2 streamed elements + the full-tile copies of the diagonal blocks (32*Np), Np = N padded to 32
    assert lib.lib.bcbf_lop_elems_f32(512) == 512 * 514 // 2 + 32 * 512
    assert lib.lib.bcbf_lop_elems_f64(256) == 256 * 258 // 2 + 32 * 256
    assert lib.lib.bcbf_lop_elems_f32(40) == 64 * 66 // 2 + 32 * 64


def test_ops_refuse_cpu_tensors(lib):
    from bayesian_cbf_amd import ops
    X = torch.zeros(1, 8, 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.kb_build(X, torch.zeros(1, 8, 2), torch.eye(2)[None], torch.ones(1, 2), torch.ones(1))


def test_lds_read_wait_lint_flags_early_use_and_accepts_waited_use():
    """bayesian_cbf_amd/check_lds_waits.py (run by the build on posterior_shared_reg.hip's device assembly): an instruction
    that touches the destination of an LDS read before the covering s_waitcnt is reported; counted waits retire the oldest
    reads first; a kernel whose name does not match is ignored."""
    from bayesian_cbf_amd.check_lds_waits import check
    good = """
_Z6kernelv:
\tds_read_b64 v[10:11], v3 offset:16
\tds_read_b32 v12, v3
\tv_add_f32_e32 v1, v2, v4
\ts_waitcnt lgkmcnt(1)
\tv_mul_f64 v[20:21], v[10:11], v[10:11]
\ts_waitcnt lgkmcnt(0)
\tv_mfma_f32_16x16x4_f32 a[0:3], v12, v12, a[0:3]
\ts_endpgm
"""
    bad, n = check(good, "kernel")
    assert n == 1 and bad == []
    early = good.replace("\tv_add_f32_e32 v1, v2, v4", "\tv_add_f32_e32 v1, v11, v4")       # reads v11 before the wait
    bad, _ = check(early, "kernel")
    assert len(bad) == 1 and bad[0][3] == [("v", 11)]
    clobber = good.replace("\tv_add_f32_e32 v1, v2, v4", "\tv_mov_b32_e32 v12, 0")            # overwrites an in-flight destination
    assert len(check(clobber, "kernel")[0]) == 1
    wrong_wait = good.replace("lgkmcnt(1)", "lgkmcnt(2)")                                     # nothing retired yet
    assert len(check(wrong_wait, "kernel")[0]) == 1
    assert check(early, "other_name") == ([], 0)
    # packed fp32: a 64-bit source pair is read only where op_sel / op_sel_hi point -- v[10:11] with op_sel_hi:[0,..] reads
    # v10 twice and never the in-flight v11; with the default modifiers it reads both halves
    pk_ok = good.replace("\tv_add_f32_e32 v1, v2, v4", "\tds_read_b32 v9, v3\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b32 v11, v3\n"
                         "\tv_pk_fma_f32 v[30:31], v[10:11], v[32:33], v[30:31] op_sel_hi:[0,1,1]\n\ts_waitcnt lgkmcnt(0)")
    assert check(pk_ok, "kernel")[0] == []
    pk_bad = pk_ok.replace(" op_sel_hi:[0,1,1]", "")
    assert [b[3] for b in check(pk_bad, "kernel")[0]] == [[("v", 11)]]
    # the compiler's 64-bit multiply expansion names an undefined high word in the addend pair: exempt only in that exact shape
    mul64 = good.replace("\tv_add_f32_e32 v1, v2, v4", "\tv_mov_b32_e32 v10, v7\n\tv_mad_u64_u32 v[30:31], s[8:9], v31, 12, v[10:11]")
    assert [b[3] for b in check(mul64, "kernel")[0]] == [[("v", 10)], [("v", 10), ("v", 11)]]   # (here v10 itself is in flight: both lines are violations)
    mul64_ok = good.replace("ds_read_b64 v[10:11]", "ds_read_b32 v11").replace(
        "\tv_add_f32_e32 v1, v2, v4", "\tv_mov_b32_e32 v10, v7\n\tv_mad_u64_u32 v[30:31], s[8:9], v31, 12, v[10:11]").replace(
        "v_mul_f64 v[20:21], v[10:11], v[10:11]", "v_mul_f32 v20, v11, v11")
    assert check(mul64_ok, "kernel")[0] == []
    assert len(check(mul64_ok.replace("\tv_mov_b32_e32 v10, v7\n", ""), "kernel")[0]) == 1       # without the preceding low-word move: reported
