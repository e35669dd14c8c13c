#!/usr/bin/env python3
"""Static check of the emitted gfx950 ISA: no register that an INLINE-ASM load is still writing may be touched before the wait
that covers the load.

Why ("an inline-asm load must be tied to its wait"): hipcc keeps no books on loads issued
from inline asm -- it inserts no s_waitcnt for their result and treats the destination register as free once its last reader is
scheduled.  The LDS / memory unit writes the register when the data returns, whatever lives there by then.  Round 2's gathered GEMM
carried a dead `ds_read_b32` whose register the compiler reused; the result was a wrong 32 x 32 block once in a few hundred launches
under load.  A probabilistic stress test found it; this finds the whole class deterministically, on the CPU box.

Method: compile a .hip source to device assembly (`hipcc -S --cuda-device-only`, same flags as the product build; inline asm is
bracketed by `;;#ASMSTART` / `;;#ASMEND` in the output), rebuild each function's control-flow graph from its labels and branches, and
run a forward data-flow over it.  The state is the set of pending asm loads: (destination registers, counter -- lgkmcnt for LDS
reads, vmcnt for global / buffer loads --, the number of later operations on the same in-order counter).  `s_waitcnt cnt(N)` retires
every pending load with at least N later operations on that counter (lgkmcnt with scalar-memory operations in flight, which return
out of order: only N = 0 retires).  Any other instruction that names a register of a pending load -- as a source (stale data) or as a
destination (clobbered when the load lands) -- is a finding.

    python tools/asm_lint.py seq-recommendations_amd/csrc/gemm.hip [more.hip ...]      exit code 1 on findings
"""
import os
import re
import subprocess
import sys
from collections import namedtuple

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]      # seq-recommendations_amd/build.py

Ins = namedtuple("Ins", "line text mnem in_asm")
Pending = namedtuple("Pending", "file lo hi kind later smem origin")       # file: 'v' | 'a'

_REG = re.compile(r"\b([va])(?:(\d+)\b|\[(\d+):(\d+)\])")
_LABEL = re.compile(r"^([.\w$]+):")
_WAIT = re.compile(r"(vmcnt|lgkmcnt)\((\d+)\)")


def device_asm(src, out=None, flags=None):
    """hipcc -S --cuda-device-only of one .hip source -> path of the .s file (cached by mtime next to the objects)."""
    src = os.path.abspath(src)
    if out is None:
        d = os.path.join(os.path.dirname(src), "_obj")
        os.makedirs(d, exist_ok=True)
        out = os.path.join(d, os.path.basename(src) + ".s")
    deps = [src] + [os.path.join(os.path.dirname(src), f) for f in os.listdir(os.path.dirname(src)) if f.endswith(".h")]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    hipcc = "/opt/rocm/bin/hipcc"
    r = subprocess.run([hipcc] + (flags or CFLAGS) + ["-S", "--cuda-device-only", src, "-o", out], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed on %s:\n%s%s" % (src, r.stdout, r.stderr))
    return out


def regs(text):
    """[(file, lo, hi)] of every VGPR / AGPR operand named in an instruction's operand text."""
    out = []
    for m in _REG.finditer(text):
        if m.group(2) is not None:
            out.append((m.group(1), int(m.group(2)), int(m.group(2))))
        else:
            out.append((m.group(1), int(m.group(3)), int(m.group(4))))
    return out


def functions(path):
    """-> [(name, [item])]; item = ('label', name, line) | Ins.  Device functions only (between a global label and .Lfunc_end)."""
    out, cur, name, in_asm = [], None, None, False
    with open(path) as f:
        for no, raw in enumerate(f, 1):
            s = raw.strip()
            if not s:
                continue
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            m = _LABEL.match(s)
            if m and not in_asm:
                lab = m.group(1)
                if lab.startswith(".Lfunc_end"):
                    if cur is not None:
                        out.append((name, cur))
                    cur = name = None
                elif not lab.startswith("."):
                    cur, name = [], lab
                elif cur is not None:
                    cur.append(("label", lab, no))
                continue
            if cur is None or s.startswith(";") or s.startswith("."):
                continue
            # inline asm may hold several instructions per line separated by newlines already; strip trailing comments
            code = s.split(";")[0].strip()
            if not code:
                continue
            mnem = code.split()[0]
            cur.append(Ins(no, code, mnem, in_asm))
    return out


def is_branch(mn):
    return mn in ("s_branch", "s_endpgm", "s_setpc_b64") or mn.startswith("s_cbranch")


def asm_load(ins):
    """(file, lo, hi, kind) if `ins` is a load with a register destination, else None."""
    mn = ins.mnem
    if mn.startswith("ds_read") or mn.startswith("ds_load") or mn in ("ds_bpermute_b32", "ds_permute_b32", "ds_swizzle_b32"):
        kind = "lgkm"
    elif re.match(r"(global|buffer|flat|scratch)_(load|atomic)", mn):
        if "_lds_" in mn or re.search(r"\blds\b", ins.text):
            return None                                   # LDS-DMA: no register destination
        if "atomic" in mn and not re.search(r"\b(glc|sc0)\b", ins.text):
            return None                                   # non-returning atomic
        kind = "vm"
    else:
        return None
    ops = ins.text[len(mn):]
    r = regs(ops.split(",")[0])
    if not r:
        return None
    f, lo, hi = r[0]
    return f, lo, hi, kind


def counter_of(mn, text):
    """Which in-order counter an instruction bumps: 'lgkm', 'smem' (lgkm, out of order), 'vm' or None."""
    if mn.startswith("ds_"):
        return "lgkm"
    if mn.startswith("s_load") or mn.startswith("s_buffer_load") or mn.startswith("s_memtime") or mn.startswith("s_memrealtime") \
            or mn.startswith("s_sendmsg") or mn.startswith("s_store") or mn.startswith("s_dcache"):
        return "smem"
    if re.match(r"(global|buffer|flat|scratch)_", mn):
        return "vm"
    return None


def lint_function(name, items, fname):
    # ---- basic blocks
    blocks, cur, labels = [], [], {}
    for it in items:
        if not isinstance(it, Ins):
            if cur:
                blocks.append(cur)
                cur = []
            labels[it[1]] = len(blocks)
            continue
        cur.append(it)
        if is_branch(it.mnem):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    # labels that pointed at an index past the end (label before nothing)
    succ = []
    for i, b in enumerate(blocks):
        s = []
        last = b[-1] if b else None
        if last is not None and is_branch(last.mnem):
            if last.mnem.startswith("s_cbranch") or last.mnem == "s_branch":
                tgt = last.text.split()[-1]
                if tgt in labels and labels[tgt] < len(blocks):
                    s.append(labels[tgt])
            if last.mnem.startswith("s_cbranch") and i + 1 < len(blocks):
                s.append(i + 1)
        elif i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    # fix: a label recorded when `blocks` was shorter points at the block that STARTS there -- true by construction above
    state_in = [dict() for _ in blocks]          # origin line -> Pending
    findings, first = {}, {}        # one finding per pending load: the first instruction that touches it

    def merge(dst, src):
        changed = False
        for k, p in src.items():
            q = dst.get(k)
            if q is None:
                dst[k] = p
                changed = True
            else:
                n = q._replace(later=min(q.later, p.later), smem=q.smem or p.smem)
                if n != q:
                    dst[k] = n
                    changed = True
        return changed

    def run_block(i, report):
        st = dict(state_in[i])
        for ins in blocks[i]:
            ld = asm_load(ins) if ins.in_asm else None
            # 1. does the instruction touch a register of a pending load?
            if st:
                named = regs(ins.text[len(ins.mnem):])
                for k, p in list(st.items()):
                    if ld is not None and ins.line == k:
                        continue                          # the load itself, re-issued on a loop path
                    for f, lo, hi in named:
                        if f == p.file and lo <= p.hi and hi >= p.lo:
                            if report and (k not in first or ins.line < first[k]):
                                first[k] = ins.line
                                findings[k] = ("%s: %s line %d `%s` touches %s[%d:%d] of the inline-asm load at line %d (`%s`) "
                                                    "before a covering s_waitcnt %scnt" % (fname, name[:60], ins.line, ins.text, p.file, p.lo, p.hi, k,
                                                                                          p.origin, "lgkm" if p.kind == "lgkm" else "vm"))
                            break
            # 2. waits retire
            if ins.mnem == "s_waitcnt":
                for cnt, n in _WAIT.findall(ins.text):
                    n = int(n)
                    kind = "lgkm" if cnt == "lgkmcnt" else "vm"
                    for k, p in list(st.items()):
                        if p.kind != kind:
                            continue
                        if n == 0 or (p.later >= n and not (kind == "lgkm" and p.smem)):
                            del st[k]
                if not _WAIT.search(ins.text) and re.search(r"s_waitcnt\s+(0|0x0)\b", ins.text):
                    st.clear()
            # 3. later operations on the in-order counters
            c = counter_of(ins.mnem, ins.text)
            if c is not None and st:
                for k, p in list(st.items()):
                    if c == "smem" and p.kind == "lgkm":
                        st[k] = p._replace(smem=True)
                    elif c == p.kind:
                        st[k] = p._replace(later=p.later + 1)
            # 4. a new pending asm load
            if ld is not None:
                f, lo, hi, kind = ld
                st[ins.line] = Pending(f, lo, hi, kind, 0, False, ins.text)
        return st

    work = list(range(len(blocks)))
    guard = 0
    while work:
        guard += 1
        if guard > 200000:
            raise RuntimeError("asm_lint: data-flow did not converge in %s" % name)
        i = work.pop()
        out = run_block(i, report=False)
        for j in succ[i]:
            if merge(state_in[j], out) and j not in work:
                work.append(j)
    for i in range(len(blocks)):
        run_block(i, report=True)
    return [findings[k] for k in sorted(findings)]


def lint_asm(path, fname=None):
    """-> (findings, stats) for one .s file."""
    out, n_fn, n_loads = [], 0, 0
    for name, items in functions(path):
        n_fn += 1
        n_loads += sum(1 for it in items if isinstance(it, Ins) and it.in_asm and asm_load(it) is not None)
        out += lint_function(name, items, fname or os.path.basename(path))
    return out, {"functions": n_fn, "asm_loads": n_loads}


def lint_source(src, out=None, flags=None):
    return lint_asm(device_asm(src, out=out, flags=flags), os.path.basename(src))


def main(argv):
    bad = 0
    for src in argv:
        f, st = lint_source(src) if not src.endswith(".s") else lint_asm(src)
        print("%s: %d device functions, %d inline-asm loads with a register destination, %d finding(s)" % (src, st["functions"], st["asm_loads"], len(f)))
        for x in f[:40]:
            print("  " + x)
        bad += len(f)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
