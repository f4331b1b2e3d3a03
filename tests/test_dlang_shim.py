"""Static cross-check of the D half of the drop-in (dlang/source/mir/optim/*.d) against the C header and
the built library. There is no D compiler in the image, so these files have never met one; what CAN be checked without one is
checked here, on the CPU tier:

  * the sources are lexically sound (comments, strings, balanced (), [], {});
  * every `extern(C)` prototype they declare names a symbol that include/mir_optim_amd.h declares AND libmir_optim_amd.so exports;
  * the status enums carry the values of the C header's (reference least_squares.d:20-46, boxcqp.d:18-26);
  * the PODs declare the fields of the C structs, in their order, with matching types (LS:85-143, QP:56-71);
  * the D default initialisers of LeastSquaresSettings!T / BoxQPSettings!T evaluate to what mir_least_squares_init_d/_s -- the
    library's own defaults, pinned against the reference's in test_abi.py -- write, for double and float;
  * the `static assert`ed sizes in the D files are the sizes of the ctypes mirrors of the C structs.
"""
import ctypes as C
import math
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DDIR = os.path.join(ROOT, "dlang", "source", "mir", "optim")
HEADER = open(os.path.join(ROOT, "include", "mir_optim_amd.h")).read()


def d_source(name):
    return open(os.path.join(DDIR, name)).read()


def strip_d(src):
    """D source without comments (`//`, `/* */`, nesting `/+ +/`) and with string / character literals emptied."""
    out, i, n = [], 0, len(src)
    while i < n:
        two = src[i:i + 2]
        if two == "//":
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif two == "/*":
            j = src.find("*/", i + 2)
            assert j >= 0, "unterminated /* comment"
            i = j + 2
        elif two == "/+":
            depth, i = 1, i + 2
            while depth:
                assert i < n, "unterminated /+ comment"
                if src[i:i + 2] == "/+":
                    depth, i = depth + 1, i + 2
                elif src[i:i + 2] == "+/":
                    depth, i = depth - 1, i + 2
                else:
                    i += 1
        elif src[i] == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('""')
            i = j + 1
        elif src[i] == "`":
            j = src.find("`", i + 1)
            assert j >= 0
            out.append('""')
            i = j + 1
        elif src[i] == "'" and re.match(r"'(\\.|[^'\\])'", src[i:i + 4]):
            m = re.match(r"'(\\.|[^'\\])'", src[i:i + 4])
            out.append("' '")
            i += m.end()
        else:
            out.append(src[i])
            i += 1
    return "".join(out)


def test_d_sources_are_lexically_sound():
    for name in ("least_squares.d", "boxcqp.d"):
        code = strip_d(d_source(name))
        stack = []
        pairs = {")": "(", "]": "[", "}": "{"}
        for pos, ch in enumerate(code):
            if ch in "([{":
                stack.append((ch, pos))
            elif ch in ")]}":
                assert stack and stack[-1][0] == pairs[ch], (name, ch, code[max(0, pos - 60):pos + 20])
                stack.pop()
        assert not stack, (name, stack[-1])
        assert re.search(r"^module mir\.optim\.\w+;", code, re.M) and "version (mir_optim_amd):" in code


def d_extern_c_functions(code):
    """Names of the functions declared inside `extern(C) ... { ... }` blocks (prototypes only: no bodies there)."""
    names = []
    for m in re.finditer(r"extern\(C\)[^{;=]*\{", code):
        depth, j = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(code[j], 0)
            j += 1
        block = code[m.end():j - 1]
        names += re.findall(r"\b(mir_\w+)\s*\(", block)
    return names


def test_every_extern_c_prototype_is_declared_by_the_header_and_exported_by_the_library():
    from mir_optim_amd import api
    L = api.lib()
    seen = []
    for name in ("least_squares.d", "boxcqp.d"):
        seen += d_extern_c_functions(strip_d(d_source(name)))
    # the reference's eleven symbols (LS:637-799, QP:36-50) + the standalone QP entries the boxcqp shim forwards to
    assert set(seen) >= {"mir_optimize_least_squares_d", "mir_optimize_least_squares_s", "mir_least_squares_work_length",
                         "mir_least_squares_iwork_length", "mir_least_squares_status_string", "mir_least_squares_init_d",
                         "mir_least_squares_init_s", "mir_least_squares_reset_d", "mir_least_squares_reset_s",
                         "mir_box_qp_work_length", "mir_box_qp_iwork_length"}
    for sym in seen:
        assert re.search(r"\b%s\s*\(" % re.escape(sym), HEADER), f"{sym}: not declared in include/mir_optim_amd.h"
        assert hasattr(L, sym), f"{sym}: not exported by libmir_optim_amd.so"


def d_enum(code, name):
    m = re.search(r"enum\s+%s\s*:\s*int\s*\{(.*?)\}" % name, code, re.S)
    assert m, name
    return {k: int(v) for k, v in re.findall(r"(\w+)\s*=\s*(-?\d+)", m.group(1))}


def c_enum(prefix):
    return {k: int(v) for k, v in re.findall(r"\b%s(\w+)\s*=\s*(-?\d+)" % prefix, HEADER)}


def test_status_enums_carry_the_c_values():
    ls = d_enum(strip_d(d_source("least_squares.d")), "LeastSquaresStatus")
    assert ls == c_enum("mir_ls_") and len(ls) == 12 and ls["numericError"] == -26
    qp = d_enum(strip_d(d_source("boxcqp.d")), "BoxQPStatus")
    assert qp == c_enum("mir_box_qp_(?!work|iwork)") and qp == {"solved": 0, "numericError": 1, "maxIterations": 2}


def d_struct_fields(code, name):
    """[(type, field, initialiser or None)] of `struct name(T) ... { ... }` (data members only)."""
    m = re.search(r"struct\s+%s\s*\(T\)[^{]*\{" % name, code)
    assert m, name
    depth, j = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(code[j], 0)
        j += 1
    fields = []
    for stmt in code[m.end():j - 1].split(";"):
        stmt = " ".join(stmt.split())
        if not stmt or stmt.startswith("import "):
            continue
        fm = re.match(r"([\w!.]+)\s+(\w+)(?:\s*=\s*(.+))?$", stmt)
        assert fm, (name, stmt)
        fields.append(fm.groups())
    return fields


def c_struct_fields(name):
    m = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), HEADER, re.S)
    assert m, name
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    fields = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        ty = stmt.split(" ", 1)[0]
        for f in stmt.split(" ", 1)[1].split(","):
            fields.append((ty, f.strip()))
    return fields


def test_pods_declare_the_fields_of_the_c_structs_in_order():
    ls = strip_d(d_source("least_squares.d"))
    qp = strip_d(d_source("boxcqp.d"))
    ctype = {"uint": "uint32_t", "T": None, "LeastSquaresStatus": "int32_t", "BoxQPSettings!T": "mir_box_qp_settings_"}
    for dname, cbase, code in (("LeastSquaresSettings", "mir_least_squares_settings_", ls),
                               ("LeastSquaresResult", "mir_least_squares_result_", ls), ("BoxQPSettings", "mir_box_qp_settings_", qp)):
        df = d_struct_fields(code, dname)
        for suffix, real in (("d", "double"), ("s", "float")):
            cf = c_struct_fields(cbase + suffix)
            assert [f for _, f, _ in df] == [f for _, f in cf], (dname, suffix)
            for (dt, fname, _), (ct, _) in zip(df, cf):
                want = {"T": real, "BoxQPSettings!T": "mir_box_qp_settings_" + suffix}.get(dt, ctype.get(dt))
                assert want == ct, (dname, fname, dt, ct)


def eval_d_default(expr, real):
    """The value of a D default initialiser of the settings structs for T = real (the handful of forms the shim uses)."""
    fi = np.finfo(real)
    mant = 53 if real is np.float64 else 24
    if expr is None:
        return 0
    e = expr.strip()
    e = e.replace("((1 - T.mant_dig) / 2)", str(int((1 - mant) / 2)))           # D integer division truncates: -26, -11 (quirk Q10)
    e = e.replace("T.max.sqrt", "math.sqrt(T_max)").replace("T.epsilon", "T_eps").replace("T.min_normal", "T_min").replace("T.max", "T_max")
    e = e.replace("T(2)", "2.0").replace("^^", "**").replace("GoldenRatio", "((1 + math.sqrt(5.0)) / 2)")
    assert re.fullmatch(r"[\w\s.+\-*/()]+", e), expr
    return eval(e, {"math": math, "T_max": float(fi.max), "T_eps": float(fi.eps), "T_min": float(fi.tiny)})


def test_d_default_initialisers_are_the_library_defaults():
    from mir_optim_amd import api
    ls = d_struct_fields(strip_d(d_source("least_squares.d")), "LeastSquaresSettings")
    qp = d_struct_fields(strip_d(d_source("boxcqp.d")), "BoxQPSettings")
    for real in (np.float64, np.float32):
        s = api.LeastSquaresSettings(real)                        # mir_least_squares_init_d/_s wrote it
        for ty, name, init in ls:
            if ty == "BoxQPSettings!T":
                for qty, qname, qinit in qp:
                    got, want = getattr(s.qpSettings, qname), eval_d_default(qinit, real)
                    assert got == (real(want) if qty == "T" else want), ("qpSettings." + qname, got, want)
                continue
            got, want = getattr(s, name), eval_d_default(init, real)
            if ty == "T":
                # (the double expression rounded to T: the D compiler folds in real precision, the values here are exact in both)
                assert got == real(want) or math.isclose(got, want, rel_tol=float(np.finfo(real).eps)), (name, got, want)
            else:
                assert got == want, (name, got, want)
    assert eval_d_default("T(2) ^^ ((1 - T.mant_dig) / 2)", np.float64) == 2.0 ** -26
    assert eval_d_default("T(2) ^^ ((1 - T.mant_dig) / 2)", np.float32) == 2.0 ** -11       # quirk Q10


def test_static_asserted_sizes_are_the_c_sizes():
    from mir_optim_amd import api
    code = strip_d(d_source("least_squares.d")) + strip_d(d_source("boxcqp.d"))
    sizes = {(t, r): int(v) for t, r, v in re.findall(r"(\w+)!(double|float)\.sizeof\s*==\s*(\d+)", code)}
    assert sizes[("LeastSquaresSettings", "double")] == C.sizeof(api.LeastSquaresSettings(np.float64)) == 128
    assert sizes[("LeastSquaresSettings", "float")] == C.sizeof(api.LeastSquaresSettings(np.float32)) == 68
    assert sizes[("LeastSquaresResult", "double")] == 32 and sizes[("LeastSquaresResult", "float")] == 24
    assert sizes[("BoxQPSettings", "double")] == 24 and sizes[("BoxQPSettings", "float")] == 12
    offs = {r: int(v) for r, v in re.findall(r"LeastSquaresSettings!(double|float)\.qpSettings\.offsetof\s*==\s*(\d+)", code)}
    assert offs == {"double": 104, "float": 56}
    assert type(api.LeastSquaresSettings(np.float64)).qpSettings.offset == 104
    assert type(api.LeastSquaresSettings(np.float32)).qpSettings.offset == 56
