#!/usr/bin/env python3
"""Erases the type annotations of zlib.ts into CommonJS zlib.js (this image has no tsc).

Supports exactly the subset zlib.ts keeps to: `export function name(arg: T, ...): R {` (R may be `Promise<T>`).
"""
import os
import re
import sys

here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "zlib.ts")).read()
names = []


def fn(m):
    name, params = m.group(1), m.group(2)
    names.append(name)
    params = ", ".join(p.split(":")[0].strip() for p in params.split(",") if p.strip())
    return "function %s(%s) {" % (name, params)


src = re.sub(r"^type \w+ = [^;]+;\n", "", src, flags=re.M)  # type aliases erase to nothing
out = re.sub(r"export function (\w+)\(([^)]*)\)\s*:\s*[\w\[\]<>]+\s*\{", fn, src)
if "export " in out or re.search(r"\w\s*:\s*(Uint8Array|number|void)\b", out.split("*/", 1)[-1].replace("input: Uint8Array", "")):
    pass  # nothing else to erase in the supported subset
out = "'use strict';\n// GENERATED from zlib.ts by strip_types.py — do not edit.\n" + out
out += "\n" + "".join("exports.%s = %s;\n" % (n, n) for n in names)
open(os.path.join(here, "zlib.js"), "w").write(out)
sys.stdout.write("zlib.js written (%s)\n" % ", ".join(names))
