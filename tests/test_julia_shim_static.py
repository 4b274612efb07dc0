"""Static check of the Julia shim (bindings/julia/JuliaGridHIP.jl).  Julia runs in neither build container, so what keeps the shim's `ccall`s
honest is this parser: every `ccall((:jg_x, lib), Ret, (T1, ...), a1, ...)` is held against the prototype of `jg_x` in include/jgrid.h --
the symbol exists, the return type and every argument type map onto the C type, the tuple of types and the arguments passed have the prototype's
arity -- and the binding table of INTEGRATION.md is held against the shim (every `jg_*` entry the table names for a Julia-side method is bound).
VERDICT r03, items 2 / 3 / 5."""
import os
import re

from conftest import ROOT

SHIM = os.path.join(ROOT, "bindings", "julia", "JuliaGridHIP.jl")
HEADER = os.path.join(ROOT, "include", "jgrid.h")


def _split_top(s):
    """comma-separated pieces of s at nesting depth 0 (parentheses, brackets, braces; string literals skipped)"""
    out, depth, cur, i = [], 0, [], 0
    while i < len(s):
        ch = s[i]
        if ch == '"':
            j = s.index('"', i + 1)
            cur.append(s[i:j + 1]); i = j + 1
            continue
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append("".join(cur).strip()); cur = []
        else:
            cur.append(ch)
        i += 1
    if "".join(cur).strip():
        out.append("".join(cur).strip())
    return out


def _ccalls(text):
    """(symbol, return type, [argument types], number of arguments passed, line) of every ccall"""
    text = re.sub(r"#[^\n]*", "", text)                           # comments (the shim has no '#' inside string literals on ccall lines)
    calls = []
    for m in re.finditer(r"\bccall\(", text):
        i, depth = m.end(), 1
        while depth:
            if text[i] == '"':
                i = text.index('"', i + 1)
            elif text[i] in "([{":
                depth += 1
            elif text[i] in ")]}":
                depth -= 1
            i += 1
        parts = _split_top(text[m.end():i - 1])
        sym = re.fullmatch(r"\(\s*:(\w+)\s*,\s*lib\s*\)", parts[0])
        assert sym, parts[0]
        types = parts[2].strip()
        assert types.startswith("(") and types.endswith(")"), types
        tl = _split_top(types[1:-1])
        calls.append((sym.group(1), parts[1].strip(), tl, len(parts) - 3, text.count("\n", 0, m.start()) + 1))
    return calls


def _prototypes(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?\w+\s*\**)\s*(jg_\w+)\s*\(([^;{]*)\)\s*;", text, flags=re.M):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        al = [] if args in ("", "void") else [re.sub(r"\s+", " ", a.strip()) for a in args.split(",")]
        protos[name] = (ret, al)
    return protos


def _ctype(arg):
    """C parameter declaration -> canonical type without the parameter name"""
    a = re.sub(r"\bconst\b", "", arg).strip()
    a = re.sub(r"\s*\*\s*", "*", a)
    m = re.fullmatch(r"(\w+)(\**)\s*(\w+)?", a) or re.fullmatch(r"(\w+)(\**)(\w+)", a)
    assert m, arg
    return m.group(1) + m.group(2)


HANDLE_TYPES = {"jg_nr", "jg_gn", "jg_comm", "jg_plan", "jg_nr_base"}
SCALARS = {"Cint": "int", "Int64": "int64_t", "Float64": "double", "Int32": "int32_t", "UInt64": "uint64_t"}
POINTEES = {"Float64": "double", "Int64": "int64_t", "Int32": "int32_t", "Int8": "int8_t", "UInt8": "uint8_t"}


def _compatible(jl, c):
    jl = jl.replace(" ", "")
    if jl in SCALARS:
        return c == SCALARS[jl]
    m = re.fullmatch(r"(?:Ptr|Ref)\{(\w+)\}", jl)
    if m:
        if m.group(1) == "Cvoid":
            return c.endswith("*") and not c.endswith("**") and (c[:-1] in HANDLE_TYPES or c[:-1] == "void")
        return c == POINTEES.get(m.group(1), "?") + "*"
    if jl in ("Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"):
        return c.endswith("**") and c[:-2] in HANDLE_TYPES
    return False


def test_every_ccall_matches_its_prototype():
    calls = _ccalls(open(SHIM).read())
    protos = _prototypes(open(HEADER).read())
    assert len(calls) >= 60 and len(protos) >= 70
    for sym, ret, types, npassed, line in calls:
        assert sym in protos, f"line {line}: {sym} is not declared in include/jgrid.h"
        cret, cargs = protos[sym]
        where = f"JuliaGridHIP.jl:{line} {sym}"
        assert len(types) == len(cargs), f"{where}: {len(types)} argument types for {len(cargs)} parameters"
        assert npassed == len(cargs), f"{where}: {npassed} arguments passed for {len(cargs)} parameters"
        if cret == "void":
            assert ret == "Cvoid", where
        elif cret == "int":
            assert ret == "Cint", where
        elif cret.replace(" ", "") == "constchar*":
            assert ret == "Cstring", where
        else:
            raise AssertionError(f"{where}: return type {cret} is not handled by the shim check")
        for k, (jl, c) in enumerate(zip(types, cargs)):
            assert _compatible(jl, _ctype(c)), f"{where}: argument {k + 1} is {jl} for `{c}`"


def test_the_shim_binds_what_the_integration_table_promises():
    """every row of the binding table that names a JuliaGrid method lists the C-ABI calls behind it; the shim must bind each of
    them (rows that describe Python-only or library-internal paths say so and are skipped)."""
    bound = {c[0] for c in _ccalls(open(SHIM).read())}
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = [l for l in md.splitlines() if l.startswith("| ") and "jg_" in l]
    assert len(table) >= 25
    missing = []
    for row in table:
        cells = [c.strip() for c in row.strip("|").split("|")]
        promised = set(re.findall(r"`(jg_\w+?)(?:/_\w+)*`", cells[1])) | set(re.findall(r"`(jg_\w+)\(", cells[1]))
        for extra in re.findall(r"`(jg_\w+)((?:/_\w+)+)`", cells[1]):                   # `jg_nr_get_jacobian/_mismatch/_increment/_maps`
            stem = extra[0].rsplit("_", 1)[0]
            promised |= {stem + suf for suf in extra[1].split("/") if suf}
        if "[python host only]" in cells[0] or "[library]" in cells[0]:
            continue
        for sym in promised:
            if sym not in bound:
                missing.append((sym, cells[0][:60]))
    assert not missing, missing
    # and the five reference entry points VERDICT r03 found unbound
    text = open(SHIM).read()
    for fn in ("fastNewtonRaphsonBX(system::PowerSystem, ::Type{HIP}", "fastNewtonRaphsonXB(system::PowerSystem, ::Type{HIP}", "function power!(analysis::HipAnyPowerFlow)",
               "function current!(analysis::HipAnyPowerFlow)", "struct HIPOrthogonal <: JuliaGrid.WlsMethod", "function pmuStateEstimation(monitoring::Measurement, ::Type{T}",
               "function chiTest(analysis::HipStateEstimation"):
        assert fn in text, fn


def test_the_shim_binds_the_whole_abi():
    """Every export of include/jgrid.h is reachable from Julia, except the device-free plan API (jg_plan_*: the CPU test-suite's view of the schedule)."""
    bound = {c[0] for c in _ccalls(open(SHIM).read())}
    protos = _prototypes(open(HEADER).read())
    unbound = sorted(set(protos) - bound - {"jg_plan_create", "jg_plan_destroy", "jg_plan_export", "jg_plan_comp_export"})
    assert not unbound, unbound


def test_docstrings_sit_directly_above_a_definition():
    """a docstring followed by another string literal (`\"...\"` then `\"\"\"...\"\"\"`) makes Julia document a string and the module fails to
    load.  Every string literal that starts a line at top level must be followed by a definition, not by another string literal."""
    lines = open(SHIM).read().splitlines()
    i = 0
    prev_doc_end = None
    while i < len(lines):
        l = lines[i]
        if l.startswith('"""'):
            j = i + 1
            while not lines[j].startswith('"""'):
                j += 1
            assert prev_doc_end is None or prev_doc_end != i - 1, f"two docstrings in a row at line {i + 1}"
            prev_doc_end = j
            i = j + 1
            continue
        if l.startswith('"') and not l.startswith('"""'):
            j = i
            while not (lines[j].rstrip().endswith('"') and (j > i or len(lines[j].rstrip()) > 1)):
                j += 1
            assert prev_doc_end is None or prev_doc_end != i - 1, f"two docstrings in a row at line {i + 1}"
            prev_doc_end = j
            i = j + 1
            continue
        if prev_doc_end is not None and prev_doc_end == i - 1:
            assert re.match(r"(function |struct |mutable struct |const |\w[\w!]*\(|@eval )", l), f"line {i + 1}: a docstring must be followed by a definition, found {l[:50]!r}"
        i += 1
