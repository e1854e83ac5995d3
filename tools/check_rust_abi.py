#!/usr/bin/env python3
"""Checks rust/raytracer-hip-sys/src/lib.rs against include/rt_abi.h, declaration for declaration.

There is no Rust toolchain in the build image, so the Rust side of the boundary ships uncompiled; this script is what
keeps it from drifting.  It parses both files and fails (exit 1, one line per finding) when
  * a constant (enum member or #define) is missing on either side or has another value,
  * a struct is missing, or its field list differs in names, order or types,
  * a struct's size / alignment (C layout rules, computed here) differs from what raytracer_amd/abi.py's ctypes mirror
    says -- the third witness, the one the GPU tests actually call through,
  * a function is missing or its return / argument types differ,
  * rust/src/gpu.rs uses an Rt* / RT_* / rt_* name that lib.rs does not declare, or names a field in a struct literal
    that the struct does not have.
usage: tools/check_rust_abi.py [--verbose]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_abi.h")
RUST = os.path.join(ROOT, "rust", "raytracer-hip-sys", "src", "lib.rs")
GPU_RS = os.path.join(ROOT, "rust", "src", "gpu.rs")

SCALARS = {  # C type -> (rust type, size, align)
    "uint8_t": ("u8", 1, 1), "int32_t": ("i32", 4, 4), "uint32_t": ("u32", 4, 4), "uint64_t": ("u64", 8, 8),
    "float": ("f32", 4, 4), "double": ("f64", 8, 8), "int": ("c_int", 4, 4), "char": ("c_char", 1, 1),
    "void": ("c_void", 0, 1),
}


def strip_c_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def c_int_value(s):
    s = s.strip().rstrip("uUlL")
    return int(s, 0)


def c_type_to_rust(ctype, structs):
    """'const RtXform*' -> '*const RtXform'; 'uint8_t*' -> '*mut u8'; 'RtContext**' -> '*mut *mut RtContext'."""
    t = ctype.replace("*", " * ").split()
    const = False
    base = None
    out = None
    for tok in t:
        if tok == "const":
            const = True
        elif tok == "*":
            out = ("*const " if const else "*mut ") + (out if out is not None else base)
            const = False
        else:
            base = SCALARS[tok][0] if tok in SCALARS else tok
    return out if out is not None else base


def parse_header():
    text = strip_c_comments(open(HEADER).read())
    consts, structs, funcs, opaque = {}, {}, {}, set()
    for m in re.finditer(r"#define\s+(RT_\w+)\s+([^\n]+)", text):
        if m.group(1) == "RT_ABI_H":
            continue
        consts[m.group(1)] = c_int_value(m.group(2))
    for m in re.finditer(r"\benum\s*\{(.*?)\}\s*;", text, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = item.split("=")
                nxt = c_int_value(val)
            else:
                name = item
            consts[name.strip()] = nxt
            nxt += 1
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            first, *rest = [d.strip() for d in decl.split(",")]
            mm = re.match(r"(.*?)(\w+)((?:\[\d+\])*)$", first)
            ctype = mm.group(1).strip()
            for name, dims in [(mm.group(2), mm.group(3))] + [re.match(r"(\w+)((?:\[\d+\])*)$", r).groups() for r in rest]:
                fields.append((name, ctype, [int(d) for d in re.findall(r"\[(\d+)\]", dims)]))
        structs[m.group(2)] = fields
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
        opaque.add(m.group(2))
    body = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    body = re.sub(r"\benum\s*\{.*?\}\s*;", " ", body, flags=re.S)
    for m in re.finditer(r"([\w\s\*]+?)\b(rt_\w+)\s*\(([^)]*)\)\s*;", body):
        ret = " ".join(m.group(1).split())
        args = []
        raw = m.group(3).strip()
        if raw and raw != "void":
            for a in raw.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?[\s\*])(\w+)$", a)
                # an unnamed parameter: the whole text is the type
                if mm and mm.group(2) not in SCALARS and mm.group(2) not in structs and mm.group(2) not in opaque:
                    args.append(mm.group(1).strip())
                else:
                    args.append(a)
        funcs[m.group(2)] = (ret, args)
    return consts, structs, funcs, opaque


def rust_field_type(ctype, dims, structs):
    t = c_type_to_rust(ctype, structs)
    for d in reversed(dims):
        t = f"[{t}; {d}]"
    return t


def layout(structs, name, memo, offsets=None):
    """(size, align) of a struct under the C rules for x86-64 / amdgcn hosts (natural alignment, pointers 8)."""
    if name in memo and offsets is None:
        return memo[name]
    off, align = 0, 1
    for fname, ctype, dims in structs[name]:
        if "*" in ctype:
            sz, al = 8, 8
        else:
            base = ctype.replace("const", "").strip()
            if base in SCALARS:
                _, sz, al = SCALARS[base]
            else:
                sz, al = layout(structs, base, memo)
        n = 1
        for d in dims:
            n *= d
        off = (off + al - 1) // al * al
        if offsets is not None:
            offsets.append((fname, off))
        off += sz * n
        align = max(align, al)
    memo[name] = ((off + align - 1) // align * align, align)
    return memo[name]


def parse_rust():
    text = re.sub(r"//[^\n]*", " ", open(RUST).read())
    consts, structs, funcs, opaque = {}, {}, {}, set()
    for m in re.finditer(r"pub const (\w+): (\w+) = (-?[\w]+);", text):
        consts[m.group(1)] = int(m.group(3).replace("_", ""), 0)
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*pub struct (\w+)\s*\{(.*?)\n\}", text, flags=re.S):
        fields = []
        for fm in re.finditer(r"(pub\s+)?(\w+)\s*:\s*([^\n]+?),\s*(?=\n|$)", m.group(2)):
            fields.append((fm.group(2), " ".join(fm.group(3).split()), bool(fm.group(1))))
        if [f[0] for f in fields] == ["_private"]:
            opaque.add(m.group(1))
        else:
            structs[m.group(1)] = fields
    for m in re.finditer(r"pub struct (\w+)", text):
        if m.group(1) not in structs and m.group(1) not in opaque:
            structs[m.group(1)] = None  # a struct without #[repr(C)]
    ext = re.search(r'extern "C" \{(.*)\n\}', text, flags=re.S)
    for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ext.group(1) if ext else "", flags=re.S):
        args = [" ".join(a.split(":", 1)[1].split()) for a in m.group(2).split(",") if a.strip()]
        funcs[m.group(1)] = ((m.group(3) or "()").strip(), args)
    return consts, structs, funcs, opaque


def check_gpu_rs(rconsts, rstructs, rfuncs, ropaque, problems):
    if not os.path.exists(GPU_RS):
        problems.append("rust/src/gpu.rs is missing")
        return
    text = re.sub(r"//[^\n]*", " ", open(GPU_RS).read())
    declared = set(rconsts) | set(rstructs) | set(rfuncs) | set(ropaque)
    for name in sorted(set(re.findall(r"\b(RT_[A-Z0-9_]+|Rt[A-Z]\w+|rt_[a-z_0-9]+)\b", text))):
        if name not in declared and name not in ("rt_camera", "RT_HIP_DEVICE", "RT_HIP_DEVICES"):  # a local fn / environment variables
            problems.append(f"gpu.rs uses {name}, which lib.rs does not declare")
    # struct literals `RtFoo { field: ..., .. }`: every named field exists
    for m in re.finditer(r"\b(Rt[A-Z]\w+)\s*\{", text):
        name = m.group(1)
        if not rstructs.get(name):
            continue
        before = text[max(0, m.start() - 12):m.start()]
        if re.search(r"(struct|impl|for)\s+$", before) or re.search(r"->\s*$", before):
            continue
        depth, i, start = 0, m.end() - 1, m.end()
        while i < len(text):
            if text[i] in "{([":
                depth += 1
            elif text[i] in "})]":
                depth -= 1
                if depth == 0:
                    break
            i += 1
        body, flat, d = text[start:i], [], 0
        for ch in body:  # keep only depth-0 text of the literal
            if ch in "{([":
                d += 1
            elif ch in "})]":
                d -= 1
            elif d == 0:
                flat.append(ch)
        names = {f[0] for f in rstructs[name]}
        for item in "".join(flat).split(","):
            item = item.strip()
            if not item or item.startswith(".."):
                continue
            field = item.split(":")[0].strip()
            if re.fullmatch(r"\w+", field) and field not in names:
                problems.append(f"gpu.rs: {name} literal names field `{field}`, which the struct does not have")


def main():
    verbose = "--verbose" in sys.argv
    problems = []
    cconsts, cstructs, cfuncs, copaque = parse_header()
    rconsts, rstructs, rfuncs, ropaque = parse_rust()

    for name, val in cconsts.items():
        if name not in rconsts:
            problems.append(f"constant {name} = {val} is missing in lib.rs")
        elif rconsts[name] != val and (rconsts[name] & 0xFFFFFFFF) != (val & 0xFFFFFFFF):
            problems.append(f"constant {name}: header {val}, lib.rs {rconsts[name]}")
    for name in rconsts:
        if name not in cconsts:
            problems.append(f"constant {name} is in lib.rs but not in rt_abi.h")

    for name, fields in cstructs.items():
        if rstructs.get(name) is None:
            problems.append(f"struct {name} is missing in lib.rs (or lacks #[repr(C)])")
            continue
        want = [(n, rust_field_type(t, dims, cstructs)) for n, t, dims in fields]
        got = [(n, t) for n, t, _ in rstructs[name]]
        if want != got:
            for i in range(max(len(want), len(got))):
                w = want[i] if i < len(want) else None
                g = got[i] if i < len(got) else None
                if w != g:
                    problems.append(f"struct {name} field {i}: header {w}, lib.rs {g}")
        for n, _, public in rstructs[name]:
            if not public:
                problems.append(f"struct {name}: field {n} is not pub")
    for name in rstructs:
        if name not in cstructs:
            problems.append(f"struct {name} is in lib.rs but not in rt_abi.h")
    if copaque != ropaque:
        problems.append(f"opaque handles differ: header {sorted(copaque)}, lib.rs {sorted(ropaque)}")

    # sizes: C layout rules here vs the ctypes mirror the tests call through
    sys.path.insert(0, ROOT)
    import ctypes
    from raytracer_amd import abi
    memo = {}
    records = {"RtXform": "XFORM", "RtShape": "SHAPE", "RtTriGeom": "TRI_GEOM", "RtTriNormals": "TRI_NORMALS",
               "RtNode": "NODE", "RtCsgNode": "CSG_NODE", "RtCsgRoot": "CSG_ROOT", "RtMaterial": "MATERIAL",
               "RtPattern": "PATTERN", "RtUvPattern": "UVPATTERN", "RtImage": "IMAGE", "RtLight": "LIGHT"}
    for name in cstructs:
        offsets = []
        size, align = layout(cstructs, name, memo, offsets)
        if name in records:  # the array records are numpy dtypes in abi.py: same size, same fields at the same offsets
            dt = getattr(abi, records[name], None)
            if dt is None:
                problems.append(f"struct {name} has no numpy mirror {records[name]} in raytracer_amd/abi.py")
            else:
                got = [(n, dt.fields[n][1]) for n in dt.names]
                if dt.itemsize != size or got != offsets:
                    problems.append(f"struct {name}: C layout {size} B {offsets}, abi.py {dt.itemsize} B {got}")
                elif verbose:
                    print(f"{name:16s} {size:4d} B  align {align}  {len(cstructs[name])} fields (numpy {records[name]})")
            continue
        ct = getattr(abi, name, None)
        if ct is None:
            problems.append(f"struct {name} has no ctypes mirror in raytracer_amd/abi.py")
        elif (ctypes.sizeof(ct), ctypes.alignment(ct)) != (size, align):
            problems.append(f"struct {name}: C layout {size} B align {align}, abi.py {ctypes.sizeof(ct)} B align "
                            f"{ctypes.alignment(ct)}")
        elif verbose:
            print(f"{name:16s} {size:4d} B  align {align}  {len(cstructs[name])} fields")

    for name, (ret, args) in cfuncs.items():
        if name not in rfuncs:
            problems.append(f"function {name} is missing in lib.rs")
            continue
        want_ret = "()" if ret == "void" else c_type_to_rust(ret, cstructs)
        want_args = [c_type_to_rust(a, cstructs) for a in args]
        got_ret, got_args = rfuncs[name]
        if want_ret != got_ret:
            problems.append(f"function {name}: returns {want_ret} in the header, {got_ret} in lib.rs")
        if want_args != got_args:
            problems.append(f"function {name}: arguments {want_args} in the header, {got_args} in lib.rs")
    for name in rfuncs:
        if name not in cfuncs:
            problems.append(f"function {name} is in lib.rs but not in rt_abi.h")
    for name in abi.EXPORTS:
        if name not in rfuncs:
            problems.append(f"abi.py exports {name}, lib.rs does not declare it")

    check_gpu_rs(rconsts, rstructs, rfuncs, ropaque, problems)

    for p in problems:
        print("MISMATCH:", p)
    if verbose or not problems:
        print(f"checked {len(cconsts)} constants, {len(cstructs)} structs, {len(cfuncs)} functions: "
              f"{'OK' if not problems else str(len(problems)) + ' problems'}")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
