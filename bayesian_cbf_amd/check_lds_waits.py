"""Build-time lint for kernels that issue LDS reads and wait for them by hand (posterior_shared_reg.hip: `ds_read` in
one inline-asm statement, `s_waitcnt lgkmcnt(0)` in another, tens of instructions later).

The compiler does not track the LGKM counter of inline assembly, so nothing but the source order keeps an instruction
from touching a `ds_read` destination before its data has arrived: a copy, a coalescing move, a re-materialisation or
a spill inserted between the read and the wait would read (or overwrite) stale registers, silently.  This check walks
the DEVICE ASSEMBLY of every kernel whose name matches and verifies, instruction by instruction, that no instruction
names a register that is the destination of an LDS read still in flight:

  * `ds_read*` adds its destination registers to the in-flight set, in issue order;
  * every other LGKM operation (LDS writes, scalar loads) takes a slot of the same counter; `s_waitcnt ... lgkmcnt(N)`
    retires all but the N youngest operations (in-order return, the assumption the compiler's own counted waits make);
  * any other instruction whose operands (destination or source) overlap an in-flight destination is a violation.

One idiom is exempt: the compiler's expansion of a 64-bit multiply, `v_mov_b32 vL, x` immediately followed by
`v_mad_u64_u32 v[..], s[..], y, k, v[L:L+1]`, keeps only the LOW word of the result and leaves the high word of the addend
pair undefined -- the register allocator may name any register there, including an in-flight destination, and its
contents do not reach the value.

It also covers the compiler's own LDS reads (which the compiler waits for itself), so a clean run means: in this code
object no LDS read result is consumed early, by anyone.  Control flow is ignored (the kernels are straight-line apart
from a few counted loops whose bodies end on a wait); a label does not clear the in-flight set.

    python -m bayesian_cbf_amd.check_lds_waits file.s [kernel-name regex]
"""
import re
import sys

_REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        bank = m.group(1)
        if m.group(2) is not None:
            out.add((bank, int(m.group(2))))
        else:
            out.update((bank, i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


_PK32 = ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_mov_b32")


def _touched(op, rest):
    """Registers an instruction reads or writes.  Packed-f32 operations name 64-bit source pairs but read only the halves
    their op_sel / op_sel_hi modifiers select (low result: element op_sel[k], default 0; high result: element
    op_sel_hi[k], default 1) -- `v_pk_fma_f32 v[0:1], v[70:71], ... op_sel_hi:[0,1,1]` reads v70 twice and never v71."""
    if not op.startswith(_PK32):
        return _regs(rest)
    mods = {}
    for name in ("op_sel", "op_sel_hi"):
        m = re.search(r"\b%s:\[([01,\s]+)\]" % name, rest)
        if m:
            mods[name] = [int(v) for v in m.group(1).replace(" ", "").split(",")]
    body = re.split(r"\s+(?:op_sel|op_sel_hi|neg_lo|neg_hi|clamp)\b", rest)[0]
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", body)]
    out = _regs(ops[0])                                   # the destination pair is written whole
    for k, src in enumerate(ops[1:]):
        m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", src)
        if not m:
            out |= _regs(src)                             # a single register, a constant: as written
            continue
        lo = int(m.group(2))
        sel = mods.get("op_sel", [0, 0, 0])
        sel_hi = mods.get("op_sel_hi", [1, 1, 1])
        for e in {sel[k] if k < len(sel) else 0, sel_hi[k] if k < len(sel_hi) else 1}:
            out.add((m.group(1), lo + e))
    return out


def check(asm_text, name_regex=r"."):
    """Returns a list of (kernel, line number, instruction, registers) violations."""
    pat = re.compile(name_regex)
    bad, kernel, inflight = [], None, []           # inflight: list of register sets, oldest first
    checked = 0
    prev = ""
    for ln, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.split(";")[0].strip()
        prev_line, prev = prev, (line if line and not line.startswith(".") else prev)
        if not line or line.startswith("."):
            if raw.lstrip().startswith(".end_amdhsa_kernel") or raw.lstrip().startswith(".Lfunc_end"):
                kernel, inflight = None, []
            continue
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m:
            if not m.group(1).startswith(".L") and not m.group(1).startswith("BB"):
                kernel = m.group(1) if pat.search(m.group(1)) else None
                inflight = []
                checked += kernel is not None
            continue
        if kernel is None:
            continue
        op, _, rest = line.partition(" ")
        if op.startswith("ds_read") or op.startswith("ds_load"):
            dest = rest.split(",")[0]
            touched = _regs(rest.split(",", 1)[1]) if "," in rest else set()      # the address register must be ready too
            pend = set().union(*inflight) if inflight else set()
            if touched & pend:
                bad.append((kernel, ln, line, sorted(touched & pend)))
            inflight.append(_regs(dest))
            continue
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_scratch_load"):
            # every other LGKM operation (LDS writes / atomics, scalar loads) takes a slot of the counter too: a later
            # `lgkmcnt(N)` counts it among the N youngest
            if inflight:
                hit = _regs(rest) & set().union(*inflight)
                if hit and op.startswith("ds_"):
                    bad.append((kernel, ln, line, sorted(hit)))
            inflight.append(set())
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                keep = int(m.group(1))
                inflight = inflight[len(inflight) - keep:] if keep else []
            elif re.fullmatch(r"\s*(0x[0-9a-fA-F]+|\d+)\s*", rest):             # raw immediate: treat as a full wait
                inflight = []
            continue
        if not inflight:
            continue
        pend = set().union(*inflight)
        hit = _touched(op, rest) & pend
        if hit and op == "v_mad_u64_u32":
            # 64-bit multiply expansion: undefined high word of the addend pair (see the module docstring)
            m = re.search(r",\s*v\[(\d+):(\d+)\]\s*$", rest)
            pm = re.match(r"v_mov_b32(?:_e32)?\s+v(\d+),", prev_line)
            if m and pm and int(pm.group(1)) == int(m.group(1)) and hit == {("v", int(m.group(2)))}:
                hit = set()
        if hit:
            bad.append((kernel, ln, line, sorted(hit)))
    return bad, checked


def main():
    text = open(sys.argv[1]).read()
    bad, checked = check(text, sys.argv[2] if len(sys.argv) > 2 else r".")
    for k, ln, ins, regs in bad[:40]:
        print("%s:%d: %s   <- in-flight LDS destination %s" % (k[:60], ln, ins, regs[:6]))
    print("%d kernel(s) checked, %d violation(s)" % (checked, len(bad)))
    return 1 if bad or not checked else 0


if __name__ == "__main__":
    sys.exit(main())
