#!/usr/bin/env python3
"""Static check of the hand-placed loads (csrc/jg_engine.hpp: gload16 / gload8) in the compiled ISA.

A request written as inline assembly is invisible to the compiler's wait-count bookkeeping: between the `global_load` and the hand-written
`s_waitcnt` that covers it, the destination registers hold stale data, and the compiler -- which believes the value arrived with the asm statement -- is
free to copy or use them there (it did, once: a v_mov of a just-requested register into a register tuple at a control-flow merge).  This tool walks the
control-flow graph of every kernel in a `hipcc -S` listing and reports each compiler-generated instruction that touches the destination of an asm
load while that load may still be in flight:

    state   = {register -> number of vector-memory operations issued after its load}   (minimum over paths)
    s_waitcnt vmcnt(k)  completes every load with at least k younger operations (loads and stores share the counter and complete in order)

Also reported: an asm load whose ADDRESS register is the destination of a load still in flight (an output of a multi-instruction statement that
was not declared early-clobber may share a register with an input a later instruction of the statement reads), and a block of instructions the walk never
reaches (the checker would pass vacuously for it).  Numeric local labels of GNU as inside an asm statement (`1:` ... `s_cbranch_scc1 1f`) are resolved
(`Nf` = the next definition of N, `Nb` = the previous one); a branch whose target is not found falls through conservatively.

    python tools/check_asm_loads.py file.s [...]          exit code 1 on a finding
"""
import re
import sys

VMEM = re.compile(r"^\s*(global_load|global_store|global_atomic|scratch_load|scratch_store|buffer_load|buffer_store|buffer_atomic|flat_load|flat_store|flat_atomic)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LABEL = re.compile(r"^([.\w$]+):")
BRANCH = re.compile(r"^\s*s_(cbranch_\w+|branch)\s+([.\w$]+)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernels(lines):
    """(name, [lines]) of every kernel (a function body up to its .Lfunc_end label that holds an s_endpgm)"""
    out, cur, name = [], None, None
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m:                                                     # (a body that never reached s_endpgm was not a kernel)
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if re.match(r"^\.Lfunc_end\d+:", ln):               # (a kernel has several s_endpgm: every early return is one)
                if any(re.match(r"^\s*s_endpgm", x) for x in cur):
                    out.append((name, cur))
                cur = None
                continue
            cur.append(ln)
    return out


def check_kernel(name, body):
    # instructions with an "in asm" flag
    ins, in_asm = [], False
    for ln in body:
        s = ln.split(";")[0].rstrip() if "#ASM" not in ln else ln
        if "#ASMSTART" in ln:
            in_asm = True; continue
        if "#ASMEND" in ln:
            in_asm = False; continue
        if not s.strip() or s.strip().startswith("."):
            if LABEL.match(s.strip()):
                ins.append(("label", LABEL.match(s.strip()).group(1), False))
            continue
        m = LABEL.match(s.strip())
        if m:
            ins.append(("label", m.group(1), False)); continue
        ins.append(("ins", s.strip(), in_asm))
    # basic blocks.  names: label -> block; numeric local labels (GNU as: `1:` may be defined many times) keep every definition in order
    blocks, cur, names, local_defs = [], [], {}, {}
    for kind, text, asm in ins:
        if kind == "label":
            if cur:
                blocks.append(cur); cur = []
            if text.isdigit():
                local_defs.setdefault(text, []).append(len(blocks))
            else:
                names[text] = len(blocks)
            continue
        cur.append((text, asm))
        if BRANCH.match(text) or text.startswith("s_endpgm"):
            blocks.append(cur); cur = []
    if cur:
        blocks.append(cur)

    def resolve(target, at):
        """block index of a branch target seen from block `at`; None if unknown"""
        if target in names:
            return names[target]
        m = re.fullmatch(r"(\d+)([fb])", target)
        if m and m.group(1) in local_defs:
            defs = local_defs[m.group(1)]
            if m.group(2) == "f":
                nxt = [d for d in defs if d > at]
                return nxt[0] if nxt else None
            prv = [d for d in defs if d <= at]
            return prv[-1] if prv else None
        return None

    succ = []
    for i, b in enumerate(blocks):
        last = b[-1][0] if b else ""
        m = BRANCH.match(last)
        s = []
        if m:
            tgt = resolve(m.group(2), i)
            if tgt is not None:
                s.append(tgt)
            if (m.group(1) != "branch" or tgt is None) and i + 1 < len(blocks):     # unknown target: fall through (conservative)
                s.append(i + 1)
        elif not last.startswith("s_endpgm") and i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    findings = []
    state_in = [None] * len(blocks)
    state_in[0] = {}
    work = [0]
    seen_find = set()
    while work:
        i = work.pop()
        st = dict(state_in[i])
        for text, asm in blocks[i]:
            if asm and text.startswith("global_load"):
                parts = text.split(",")
                addr = regs_of(parts[1]) if len(parts) > 1 else set()          # global_load dst, vaddr, saddr
                hit = addr & set(st)
                if hit and (i, text) not in seen_find:
                    seen_find.add((i, text))
                    findings.append((name, text + "    [address register]", sorted(hit)))
                for r in st:
                    st[r] += 1
                for r in regs_of(parts[0]):
                    st[r] = 0
                continue
            w = re.search(r"vmcnt\((\d+)\)", text) if text.startswith("s_waitcnt") else None
            if w:
                k = int(w.group(1))
                st = {r: a for r, a in st.items() if a < k}
                continue
            if not asm and st:
                hit = regs_of(text) & set(st)
                if hit and (i, text) not in seen_find:
                    seen_find.add((i, text))
                    findings.append((name, text, sorted(hit)))
            if VMEM.match(text):
                for r in st:
                    st[r] += 1
        for j in succ[i]:
            if state_in[j] is None:
                state_in[j] = dict(st); work.append(j)
            else:
                merged, changed = dict(state_in[j]), False
                for r, a in st.items():
                    if r not in merged or a < merged[r]:
                        merged[r] = a; changed = True
                if changed:
                    state_in[j] = merged; work.append(j)
    for i, b in enumerate(blocks):                              # a block the walk never reached was never analysed: no vacuous 'ok'
        if state_in[i] is None and any(not t.startswith(("s_endpgm", "s_nop", "s_code_end")) for t, _ in b):
            findings.append((name, f"unreachable block of {len(b)} instruction(s) starting at: {b[0][0]}", []))
    return findings


def main(paths):
    bad = 0
    for p in paths:
        lines = open(p).read().splitlines()
        for name, body in kernels(lines):
            if not any("#ASMSTART" in l for l in body):
                continue
            f = check_kernel(name, body)
            print(f"{p}: {name}: {'ok' if not f else str(len(f)) + ' finding(s)'}")
            for _, text, hit in f[:12]:
                print(f"    touches v{hit} while its asm load may be in flight:  {text}")
            bad += len(f)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
