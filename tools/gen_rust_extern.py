#!/usr/bin/env python
"""The Rust side of the boundary, generated: every struct, constant and function of include/summerset_hip.h as the
`#[repr(C)]` / `extern "C"` declarations a maintainer of the reference would put into `src/utils/hipshim.rs`
(INTEGRATION.md §2 shows the hand-picked ones with their call sites; §6 is this script's output, complete).

The header is plain C99 of a narrow shape -- `#define NAME literal`, anonymous `enum { .. }`, `typedef struct NAME NAME;`
(opaque handles), `typedef struct { fields } NAME;`, prototypes of `smr_*` functions over fixed-width integers, `double`,
`char`, `void`, pointers to those and to the structs, and `T name[N]` parameters -- so a few regular expressions parse it;
anything outside that shape stops the script (and `tests/test_abi.py::test_integration_md_declares_every_symbol`) instead of
being guessed at.

usage: python tools/gen_rust_extern.py            # print the block
       python tools/gen_rust_extern.py --write    # replace the block between the markers of INTEGRATION.md
       python tools/gen_rust_extern.py --check    # exit 1 if INTEGRATION.md's block is not this output
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "summerset_hip.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED: tools/gen_rust_extern.py -->", "<!-- END GENERATED: tools/gen_rust_extern.py -->"

PRIM = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int8_t": "i8", "int16_t": "i16", "int32_t": "i32",
        "int64_t": "i64", "int": "i32", "unsigned": "u32", "double": "f64", "float": "f32", "char": "c_char", "size_t": "usize", "void": "c_void"}
RUST_KEYWORDS = {"type", "ref", "in", "match", "move", "fn", "loop", "box", "self", "mod", "use", "as", "where", "impl"}


def camel(name):
    """smr_mp_tick_in -> SmrMpTickIn"""
    return "".join(p[:1].upper() + p[1:] for p in name.split("_") if p)


def strip_comments(src):
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def rust_ident(name):
    return "r#" + name if name in RUST_KEYWORDS else name


class Parser:
    def __init__(self, src):
        self.src = strip_comments(src)
        self.opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", self.src)
        self.structs = [(m.group(2), m.group(1)) for m in re.finditer(r"typedef\s+struct\s*\w*\s*\{(.*?)\}\s*(\w+)\s*;", self.src, flags=re.S)]
        self.names = set(self.opaque) | {n for n, _ in self.structs}
        self.defines = [(m.group(1), m.group(2).strip()) for m in re.finditer(r"^[ \t]*#define[ \t]+(SMR_\w+)[ \t]+(.+?)[ \t]*$", self.src, flags=re.M)]
        self.enums = []
        for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", self.src, flags=re.S):
            for item in m.group(1).split(","):
                item = item.strip()
                if item:
                    k, v = [x.strip() for x in item.split("=")]
                    self.enums.append((k, v))
        body = re.sub(r"typedef\s+struct\s*\w*\s*\{.*?\}\s*\w+\s*;", "", self.src, flags=re.S)
        body = re.sub(r"enum\s*\{.*?\}\s*;", "", body, flags=re.S)
        body = re.sub(r"^[ \t]*#.*$", "", body, flags=re.M)
        self.funcs = []
        for m in re.finditer(r"([\w\s\*]+?)\b(smr_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", body, flags=re.S):
            ret = " ".join(m.group(1).split())
            params = [" ".join(p.split()) for p in m.group(3).split(",")]
            self.funcs.append((m.group(2), ret, params))

    def base(self, t):
        if t in PRIM:
            return PRIM[t]
        if t in self.names:
            return camel(t)
        raise SystemExit("gen_rust_extern: unknown C type %r" % t)

    def ctype(self, t):
        """a C type without a declarator name -> Rust.  Handles `const T`, `T *`, `const T *`, `T *const *`, `T **`."""
        toks = re.findall(r"\w+|\*", t)
        # consume the base type (qualifiers before it)
        const_base = False
        i = 0
        while toks[i] == "const":
            const_base = True
            i += 1
        base = toks[i]
        i += 1
        if i < len(toks) and toks[i] == "const":                 # `T const`
            const_base = True
            i += 1
        ty, is_const = self.base(base), const_base
        while i < len(toks):
            assert toks[i] == "*", t
            i += 1
            ty = ("*const " if is_const else "*mut ") + ty
            is_const = False
            if i < len(toks) and toks[i] == "const":             # the pointer itself is const: the NEXT level points at const
                is_const = True
                i += 1
        return ty

    def declarator(self, p):
        """`const uint8_t *key_dev` / `uint64_t out[2]` / `const void *const send_dev[3]` -> (name, rust type)"""
        m = re.match(r"^(.*?)(\w+)\s*\[(\w+)\]$", p)
        if m:                                                    # an array parameter decays to a pointer to its element
            elem = m.group(1).strip()
            toks = re.findall(r"\w+|\*", elem)
            # `T name[N]` -> *mut T ; `const T name[N]` -> *const T ; `const void *const name[N]` -> *const *const c_void
            if toks and toks[-1] == "const" and "*" in toks:
                return m.group(2), "*const " + self.ctype(" ".join(toks[:-1]))
            if "*" not in toks and toks[0] == "const":
                return m.group(2), "*const " + self.ctype(" ".join(toks[1:]))
            return m.group(2), "*mut " + self.ctype(elem)
        m = re.match(r"^(.*?)(\w+)$", p)
        return m.group(2), self.ctype(m.group(1).strip())

    def field_lines(self, body):
        out = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"^((?:const\s+)?\w+)\s+(.*)$", decl)
            base, rest = m.group(1), m.group(2)
            for d in rest.split(","):
                d = d.strip()
                stars = d.count("*")
                name = d.replace("*", "").strip()
                arr = re.match(r"^(\w+)\s*\[(\w+)\]$", name)
                ty = self.ctype(base + " " + "*" * stars) if stars else self.ctype(base)
                if arr:
                    n = arr.group(2)
                    out.append("    pub %s: [%s; %s]," % (rust_ident(arr.group(1)), ty, n if n.isdigit() else n + " as usize"))
                else:
                    out.append("    pub %s: %s," % (rust_ident(name), ty))
        return out


def literal(v):
    v = v.strip().strip("()").strip()
    m = re.match(r"^(-?)(0x[0-9A-Fa-f]+|\d+)([uU]?)$", v)
    if not m:
        raise SystemExit("gen_rust_extern: cannot read the literal %r" % v)
    neg, num, uns = m.groups()
    return (neg + num), ("u32" if uns else "i32")                # (return codes are `int`: SMR_OK and the error classes compare as i32)


def generate():
    p = Parser(open(HEADER).read())
    out = ["// generated by tools/gen_rust_extern.py from include/summerset_hip.h -- do not edit here", "#![allow(non_upper_case_globals)]",
           "use std::os::raw::{c_char, c_void};", ""]
    for k, v in p.defines:
        if k == "SUMMERSET_HIP_H":
            continue
        lit, ty = literal(v)
        out.append("pub const %s: %s = %s;" % (k, ty, lit))
    for k, v in p.enums:
        out.append("pub const %s: i32 = %s;" % (k, v))
    out.append("")
    for n in p.opaque:
        out.append("#[repr(C)] pub struct %s { _private: [u8; 0] }" % camel(n))
    out.append("")
    for n, body in p.structs:
        out.append("#[repr(C)]")
        out.append("pub struct %s {" % camel(n))
        out += p.field_lines(body)
        out.append("}")
    out.append("")
    out.append('#[link(name = "summerset_hip")]')
    out.append('extern "C" {')
    for name, ret, params in p.funcs:
        if params == ["void"]:
            args = ""
        else:
            args = ", ".join("%s: %s" % (rust_ident(n), t) for n, t in (p.declarator(x) for x in params))
        r = "" if ret == "void" else " -> " + p.ctype(ret)
        line = "    pub fn %s(%s)%s;" % (name, args, r)
        if len(line) > 140:                                      # one parameter group per line of <= 140 columns
            import textwrap
            head = "    pub fn %s(" % name
            body = textwrap.wrap(args, 140 - 8, break_long_words=False, break_on_hyphens=False)
            line = head + ("\n" + " " * 8).join(body) + ")" + r + ";"
        out.append(line)
    out.append("}")
    return "\n".join(out) + "\n", [f[0] for f in p.funcs]


def block():
    text, _ = generate()
    return BEGIN + "\n```rust\n" + text + "```\n" + END


def main(argv):
    if "--write" in argv or "--check" in argv:
        doc = open(DOC).read()
        i, j = doc.find(BEGIN), doc.find(END)
        if i < 0 or j < 0:
            raise SystemExit("gen_rust_extern: INTEGRATION.md has no generated block (markers missing)")
        new = doc[:i] + block() + doc[j + len(END):]
        if "--check" in argv:
            if new != doc:
                raise SystemExit("gen_rust_extern: INTEGRATION.md's generated block is stale -- run tools/gen_rust_extern.py --write")
            return 0
        open(DOC, "w").write(new)
        return 0
    sys.stdout.write(generate()[0])
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
