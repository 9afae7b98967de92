"""Build libbcbf.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m bayesian_cbf_amd.build          # incremental; --force rebuilds everything

hipcc cross-compiles without a GPU.  The shared object is written next to this file so that
it travels with the source tree to the GPU box and shows up as an in-tree native library.
"""
import concurrent.futures
import os
import subprocess
import sys

if __package__ in (None, ""):                      # `python bayesian_cbf_amd/build.py`
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    __package__ = "bayesian_cbf_amd"

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libbcbf.so")
ARCH = "gfx950"
SOURCES = ["common.hip", "posterior_step.hip", "jets_mfma.hip", "posterior_shared.hip", "posterior_shared_reg.hip", "refit.hip", "refit_mfma.hip", "refit_mfma64.hip", "refit_wave64.hip", "refit_slab.hip", "solve.hip", "trtri.hip", "tail.hip", "syrk.hip", "mll_grad.hip", "fit.hip", "cbc_terms.hip", "predict_assemble.hip", "gram.hip", "controller_cones.hip", "socp.hip", "socp_quad.hip",
           "unicycle.hip", "control_step.hip"]
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
         "-I" + CSRC, "-Wall", "-Wno-unused-function", "-fvisibility=hidden"]
# per-file extras.  posterior_shared: MFMA results are consumed by VALU code every block, so keep the accumulators
# in VGPRs (no v_accvgpr round trips).  refit_wave64: the SLP vectorizer packs the K_b value pass into v_pk_mul_f32 pairs it
# has to assemble with register moves (more instructions and more live registers than the scalar fmas): without it the
# fp32 batch form at two waves per SIMD runs 4096 x 256 in 0.63 instead of 0.71 ms, everything else within 1 %
EXTRA_FLAGS = {"posterior_shared.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "refit_wave64.hip": ["-fno-slp-vectorize"],
               "posterior_shared_reg.hip": ["-Rpass-analysis=kernel-resource-usage"],
               "posterior_step.hip": ["-Rpass-analysis=kernel-resource-usage"]}
# instantiations of the streaming kernel whose measured figures assume ZERO scratch (a spill in the streaming loop is fatal
# there: DESIGN.md 3.1 -- 1373 us against 470): the headline (fp32 / fp64 values-only, C = 2..3), the fp32 jets of the pendulum and
# unicycle shapes, the fp32 fused query + append column.  Demangled-name prefixes; the build fails when one of them reports scratch
# beyond its allowance (SCRATCH_ALLOWANCE: the unicycle jets spill 40 B per lane since the block loop exists twice -- A side live /
# dead, BCBF_PS_SKIP_DEAD_A -- and run 0.444 ms with them against 0.455 without the second loop and without scratch).
ZERO_SCRATCH_KERNELS = {"posterior_step.hip": [
    "posterior_step_kernel<float, 2, 4, 0, 1, false, 0,", "posterior_step_kernel<float, 3, 4, 0, 1, false, 0,",
    "posterior_step_kernel<double, 2, 4, 0, 1, false, 0,", "posterior_step_kernel<double, 3, 4, 0, 1, false, 0,",
    "posterior_step_kernel<float, 2, 4, 2, 1, false, 0,", "posterior_step_kernel<float, 3, 4, 3, 1, false, 0,",
    "posterior_step_kernel<float, 3, 4, 0, 1, false, 1, false, false>", "posterior_step_kernel<float, 3, 4, 0, 1, true, 0,",
    "posterior_step_kernel<double, 4, 4, 0, 1, false, 0,", "posterior_step_kernel<double, 3, 8, 0, 1, false, 0,"]}
# kernels that must not touch scratch memory (posterior_shared_reg: an operand spilled between its explicit LDS read and
# the explicit wait for it would be stored before it has arrived).  Their device assembly is also linted: no instruction may
# name the destination of an LDS read that has not been waited for (check_lds_waits.py)
SCRATCH_ALLOWANCE = {"posterior_step_kernel<float, 3, 4, 3, 1, false, 0,": 40}
NO_SCRATCH = {"posterior_shared_reg.hip"}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(a, bs):
    if not os.path.exists(a):
        return False
    ta = os.path.getmtime(a)
    return all(os.path.getmtime(b) <= ta for b in bs)


# --asan: host-side AddressSanitizer build -- the launchers, argument checks and error plumbing are
# instrumented, the device code is not (-fno-gpu-sanitize: GPU ASan needs xnack+, not available on this pool).  A
# separate artefact, libbcbf_asan.so, for CPU-side runs of tests/cabi_asan_driver.c under ASan; never the product library.
ASAN_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan"]
ASAN = False


# a source compiled in several parts ("file.hip#tag" in the work list): the 27 instantiations of the register-resident regime-S
# kernel take 6 minutes in one translation unit, 1.5 in seven
PARTS = {"posterior_shared_reg.hip": dict([("base", ["-DBCBF_PSR_PART_BASE"])] +
                                          [("%s%d" % ("df"[t], c), ["-DBCBF_PSR_PART_T=%d" % t, "-DBCBF_PSR_PART_C=%d" % c])
                                           for t in (0, 1) for c in (2, 3, 4)])}


def _work_list():
    out = []
    for src in SOURCES:
        if src in PARTS and not ASAN:
            out += ["%s#%s" % (src, tag) for tag in PARTS[src]]
        else:
            out.append(src)
    return out


def _compile(item, force):
    src, _, tag = item.partition("#")
    obj = os.path.join(OBJ + ("_asan" if ASAN else ""), os.path.splitext(src)[0] + ("_" + tag if tag else "") + ".o")
    deps = [os.path.join(CSRC, src), os.path.join(ROOT, "include", "bcbf.h"), os.path.abspath(__file__)]
    deps += [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")]     # every shared header
    if not force and _newer(obj, deps):
        return obj, False
    tuning = os.environ.get("BCBF_EXTRA_HIPCC_FLAGS", "").split()        # e.g. -DBCBF_PS_UNR=2 for tuning sweeps
    flags = [f for f in FLAGS if not (ASAN and f == "-O3")] + (ASAN_FLAGS if ASAN else [])
    extra = list(EXTRA_FLAGS.get(src, [])) + (PARTS[src][tag] if tag else [])
    if ASAN and src == "posterior_shared_reg.hip":
        extra.append("-DBCBF_PSR_DEV")        # host-side instrumentation only: one device instantiation per precision is
                                              # enough there (the 18 of the product build take minutes to compile)
    cmd = [_hipcc()] + flags + extra + tuning + ["-c", os.path.join(CSRC, src), "-o", obj]
    guard = src in NO_SCRATCH and not ASAN and tag != "base"
    tmpdir = None
    if guard:
        # keep the device assembly of this compile (-save-temps, in a scratch directory) for the LDS read / wait lint below
        import tempfile
        tmpdir = tempfile.mkdtemp(prefix="bcbf_asm_")
        cmd = cmd[:-1] + [os.path.join(tmpdir, os.path.basename(obj)), "-save-temps=obj"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, res.stdout, res.stderr))
    if guard:
        import glob
        import shutil
        from .check_lds_waits import check as _check_lds
        asm = glob.glob(os.path.join(tmpdir, "*-hip-amdgcn-amd-amdhsa-%s.s" % ARCH))
        try:
            if len(asm) != 1:
                raise RuntimeError("%s: no device assembly to check (%s)" % (src, asm))
            bad, checked = _check_lds(open(asm[0]).read(), r"posterior_shared_reg_kernel")
            if bad or not checked:
                raise RuntimeError("%s [%s]: %d instruction(s) touch the destination of an LDS read that is still in flight "
                                   "(explicit ds_read / s_waitcnt pairs; first: %s)" % (src, tag, len(bad), bad[:3]))
            shutil.move(os.path.join(tmpdir, os.path.basename(obj)), obj)
        finally:
            shutil.rmtree(tmpdir, ignore_errors=True)
    if src in ZERO_SCRATCH_KERNELS and not ASAN and not tuning:
        import re
        names = re.findall(r"Function Name: (\S+)", res.stderr)
        sizes = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", res.stderr)]
        dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
        seen = set()
        for full, z in zip(dem, sizes):
            for want in ZERO_SCRATCH_KERNELS[src]:
                if want in full:
                    seen.add(want)
                    if z > SCRATCH_ALLOWANCE.get(want, 0):
                        os.remove(obj)
                        raise RuntimeError("%s: %s uses %d bytes of scratch per lane (its measured figures assume none)" % (src, want, z))
        missing = [w for w in ZERO_SCRATCH_KERNELS[src] if w not in seen]
        if missing or len(names) != len(sizes):
            os.remove(obj)          # (an unchecked object must not pass for "up to date" on the next build)
            raise RuntimeError("%s: the scratch guard found no resource remark for %s" % (src, missing or "the kernels"))
    if guard:
        import re
        names = re.findall(r"Function Name: (\S+)", res.stderr)
        sizes = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", res.stderr)]
        # (the last three template arguments are the occupancy the kernel is compiled for, its queries per wave and the data
        # kernel: only the one-wave-per-SIMD form, `Li1E`, carries the explicit read / wait pairs)
        bad = [(n_, z) for n_, z in zip(names, sizes) if z and re.search(r"ELi1ELi\dELi\dEEEv", n_)]
        if not any(re.search(r"ELi1ELi\dELi\dEEEv", n_) for n_ in names) and any("posterior_shared_reg_kernel" in n_ for n_ in names):
            raise RuntimeError("%s: no one-wave-per-SIMD kernel name matched the scratch guard (%s)" % (src, names[:2]))
        if not sizes or len(names) != len(sizes) or bad:
            os.remove(obj)
            raise RuntimeError("%s: kernels with explicit LDS read / wait pairs must not use scratch memory: %s" % (src, bad or "no resource remarks"))
    return obj, True


def build(force=False, verbose=False, asan=False):
    global ASAN
    ASAN = bool(asan)
    try:
        return _build(force, verbose)
    finally:
        ASAN = False


def _build(force, verbose):
    LIB = os.path.join(HERE, "libbcbf_asan.so") if ASAN else globals()["LIB"]
    os.makedirs(OBJ + ("_asan" if ASAN else ""), exist_ok=True)
    work = _work_list()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(work))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), work))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or not _newer(LIB, objs):
        # link to a temporary name and rename: another rank of a multi-process launch never sees a half-written library
        tmp = "%s.tmp.%d" % (LIB, os.getpid())
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
        if ASAN:
            cmd += ["-fsanitize=address", "-shared-libsan"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("link failed:\n%s\n%s" % (res.stdout, res.stderr))
        os.replace(tmp, LIB)
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, asan="--asan" in sys.argv)
