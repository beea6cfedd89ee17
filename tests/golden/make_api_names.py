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
json.dump({"defined": sorted(defs), "struct_fields": sorted(fields)}, open(os.path.join(HERE, "reference_api_names.json"), "w"), indent=0, ensure_ascii=False)
print(len(defs), "definitions,", len(fields), "struct fields")
