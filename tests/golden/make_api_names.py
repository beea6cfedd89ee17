#!/usr/bin/env python3
"""The identifiers the reference DEFINES at top level of src/*.jl (functions, structs, abstract types, consts, short-form methods) and
the field names of its structs - names only, no source text - as tests/golden/reference_api_names.json: what
tests/test_julia_ext_abi.py checks the never-executed Julia extension's `nuPGCM.<name>` references and field accesses against.
Run where /root/reference exists:  python tests/golden/make_api_names.py"""
import glob
import json
import os
import re

REF = os.environ.get("NUPGCM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ident = r"[^\W\d][\w!]*"
defs, fields = set(), set()
for f in sorted(glob.glob(os.path.join(REF, "src", "*.jl"))):
    src = open(f).read()
    for pat in (rf"\bfunction\s+(?:\w+\.)?({ident})", rf"\bstruct\s+({ident})", rf"\babstract type\s+({ident})", rf"\bconst\s+({ident})",
                rf"^\s*({ident})\s*\([^)\n]*\)\s*(?:where[^=\n]*)?=(?!=)"):
        defs.update(re.findall(pat, src, flags=re.M))
    for body in re.findall(r"\bstruct\s+[^\n]*\n(.*?)\n\s*end\b", src, flags=re.S):
        for line in body.splitlines():
            m = re.match(rf"\s*({ident})\s*(::|$)", line)
            if m and m.group(1) not in ("function", "end"):
                fields.add(m.group(1))


def arities(src, prefix):
    """name -> set of positional-argument counts of its method definitions (long and short form)"""
    out = {}
    for m in re.finditer(r"(?:^|\n)\s*(?:function\s+)?%s(%s)\s*(?:\{[^}]*\})?\(" % (prefix, ident), src):
        i, depth, args = m.end(), 1, ""
        while depth and i < len(src):
            c = src[i]
            depth += (c in "([{") - (c in ")]}")
            if depth:
                args += c
            i += 1
        d, pos = 0, ""
        for c in args:
            d += (c in "([{") - (c in ")]}")
            if c == ";" and d == 0:
                break
            pos += c
        d, n, cur = 0, 0, ""
        for c in pos:
            d += (c in "([{") - (c in ")]}")
            if c == "," and d == 0:
                n, cur = n + 1, ""
            else:
                cur += c
        out.setdefault(m.group(1), set()).add(n + (1 if cur.strip() else 0))
    return out


ar = {}
for f in sorted(glob.glob(os.path.join(REF, "src", "*.jl"))) + [os.path.join(REF, "ext", "nuPGCMCUDAExt.jl")]:
    for k, v in arities(open(f).read(), r"(?:nuPGCM\.)?").items():
        if k in defs or k in ("on_architecture", "architecture", "vector_type", "print_memory_status"):
            ar.setdefault(k, set()).update(v)
json.dump({"defined": sorted(defs), "struct_fields": sorted(fields), "arities": {k: sorted(v) for k, v in sorted(ar.items())}}, open(os.path.join(HERE, "reference_api_names.json"), "w"), indent=0, ensure_ascii=False)
print(len(defs), "definitions,", len(fields), "struct fields")
