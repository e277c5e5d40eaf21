#!/usr/bin/env python3
"""TEST INFRASTRUCTURE / build-time tool - see oracle/rc_oracle.h.

Turns the NIR listing Mesa prints for a GLSL shader (GALLIVM_DEBUG=tgsi for the vertex stage, LP_DEBUG=fs for the
fragment stage: the optimised, scalarised form llvmpipe compiles, i.e. the exact operation order of the GL the reference
is measured on) into one straight C function: one statement per NIR instruction, same order, same association.  The float
built-ins are macros (RCN_*) the including file defines - the oracle maps them to its llvmpipe-exact primitives
(o_pow, o_sin, ...), the HIP side to the device ones (pow_, sin_, ...).

    nir2c.py listing.txt --stage vertex|fragment --name fn > generated.inc

Generated signature:
    static void fn(const float* U, const float* IN, float* OUT, void* TEXCTX)
U = the default uniform block in dwords (offsets listed in fn_uniforms[]), IN = the shader inputs in declaration order
(fn_inputs[]; vec4 attributes take four consecutive floats), OUT = the outputs in declaration order (fn_outputs[]),
RCN_TEX(TEXCTX, unit, u, v, dst4) samples, RCN_TXF(TEXCTX, unit, x, y, dst4) fetches texel (x, y) of level 0.  RCN_NO_TABLES / RCN_TABLES_ONLY leave out the tables / the function.  Only what the restated shaders use is handled; anything else raises.
"""
import argparse
import re
import sys

kPosSlot = 8  # gl_Position goes after the generic varyings in OUT

ALU1 = {"fneg": "(-{0})", "fabs": "RCN_ABS({0})", "frsq": "RCN_RSQ({0})", "frcp": "RCN_RCP({0})", "fsqrt": "RCN_SQRT({0})",
        "fsign": "RCN_SIGN({0})", "fsin": "RCN_SIN({0})", "fcos": "RCN_COS({0})", "ffloor": "RCN_FLOOR({0})",
        "ffract": "RCN_FRACT({0})", "fexp2": "RCN_EXP2({0})", "flog2": "RCN_LOG2({0})", "mov": "{0}", "fsat": "RCN_SAT({0})",
        "fround_even": "RCN_RINT({0})", "ftrunc": "RCN_TRUNC({0})", "fceil": "RCN_CEIL({0})"}
ALU2 = {"fmul": "({0} * {1})", "fadd": "({0} + {1})", "fsub": "({0} - {1})", "fdiv": "RCN_DIV({0}, {1})", "fmin": "RCN_MIN({0}, {1})",
        "fmax": "RCN_MAX({0}, {1})", "fpow": "RCN_POW({0}, {1})", "fmod": "RCN_MOD({0}, {1})"}
CMP = {"flt32": "({0} < {1})", "fge32": "({0} >= {1})", "feq32": "({0} == {1})", "fneu32": "(!({0} == {1}))"}
BOOL2 = {"ior": "({0} | {1})", "iand": "({0} & {1})"}


class Gen:
    def __init__(self, name):
        self.name = name
        self.decl_f, self.decl_i, self.decl_v = set(), set(), {}
        self.lines = []
        self.ind = 1
        self.types = {}      # ssa -> 'f' | 'i' | ('v', n)
        self.deref = {}      # ssa -> variable name
        self.const = {}      # ssa -> list of raw 32-bit words
        self.inputs, self.outputs, self.uniforms = [], [], []
        self.in_off, self.out_off = {}, {}
        self.phis = {}       # pred block -> [(dst, src)]
        self.cur_block = None

    def emit(self, s):
        self.lines.append("  " * self.ind + s)

    def val(self, tok):
        """C expression of an operand token such as %70.x or %21"""
        m = re.match(r"%(\d+)(?:\.([xyzw]+))?$", tok)
        if not m:
            raise ValueError("operand " + tok)
        n, sw = m.group(1), m.group(2)
        t = self.types.get(n)
        if isinstance(t, tuple):
            if not sw or len(sw) != 1:
                raise ValueError("vector operand without a single swizzle: " + tok)
            return "v%s[%d]" % (n, "xyzw".index(sw))
        if sw and sw != "x":
            raise ValueError("swizzle on scalar " + tok)
        return ("b%s" if t == "i" else "s%s") % n

    def set(self, n, expr, ty="f"):
        self.types[n] = ty
        if ty == "i":
            self.decl_i.add(n)
            self.emit("b%s = %s;" % (n, expr))
        else:
            self.decl_f.add(n)
            self.emit("s%s = %s;" % (n, expr))

    def flush_phis(self):
        for dst, src in self.phis.get(self.cur_block, []):
            self.emit("%s = %s;" % (dst, self.val(src)))


def split_args(s):
    """operands of an ALU instruction: '%21 (0.500000), %63' -> ['%21', '%63']"""
    return re.findall(r"%\d+(?:\.[xyzw]+)?", s)


def translate(text, stage, name):
    g = Gen(name)
    lines = text.split("\n")
    # declarations
    for l in lines:
        m = re.match(r"decl_var uniform INTERP_MODE_NONE (\w+) (\w+) \((\d+), (\d+), \d+\)", l)
        if m and m.group(1) != "sampler2D":
            n = {"float": 1, "vec2": 2, "vec3": 3, "vec4": 4, "mat4": 16, "int": 1}[m.group(1)]
            g.uniforms.append((m.group(2), int(m.group(4)), n))
        m = re.match(r"decl_var shader_(in|out) (\w+) (\w+) ([\w#]+) \((\w+?)(\d*)\.([xyzw]+),", l)
        if m:
            # varyings are addressed by slot (VARYING_SLOT_VARn.c -> 4n+c) so that the two stages agree whatever the names
            n = {"float": 1, "vec2": 2, "vec3": 3, "vec4": 4}[m.group(3)]
            lst, off = (g.inputs, g.in_off) if m.group(1) == "in" else (g.outputs, g.out_off)
            if m.group(5) == "VARYING_SLOT_VAR":
                o = 4 * int(m.group(6)) + "xyzw".index(m.group(7)[0])
            elif m.group(5) == "VERT_ATTRIB_GENERIC":
                o = 4 * int(m.group(6))
            elif m.group(5) == "VARYING_SLOT_POS":
                o = 4 * kPosSlot
            elif m.group(5) == "FRAG_RESULT_DATA":
                o = 0
            else:
                raise ValueError("slot " + l)
            off[m.group(4)] = o
            lst.append((m.group(4), n, m.group(2) == "INTERP_MODE_FLAT"))
    start = next(i for i, l in enumerate(lines) if l.startswith("impl main"))
    body = []
    for l in lines[start + 1:]:
        if l.startswith("}"):
            break
        body.append(l)
    # phis first: assignments at the end of the predecessor blocks
    for l in body:
        m = re.match(r"\s*32\s+%(\d+) = phi (.*)$", l)
        if m:
            dst = m.group(1)
            for pm in re.finditer(r"(b\d+): (%\d+)", m.group(2)):
                g.phis.setdefault(pm.group(1), []).append(("PHI" + dst, pm.group(2)))
    for l in body:
        s = l.strip()
        if not s or s.startswith("//"):
            continue
        m = re.match(r"block (b\d+):", s)
        if m:
            g.cur_block = m.group(1)
            g.emit("/* %s */" % m.group(1))
            continue
        if s.startswith("if "):
            c = re.match(r"if (%\d+)", s).group(1)
            g.flush_phis()
            g.cur_block = None
            g.emit("if (%s) {" % g.val(c))
            g.ind += 1
            continue
        if s.startswith("} else {"):
            g.flush_phis()
            g.cur_block = None
            g.ind -= 1
            g.emit("} else {")
            g.ind += 1
            continue
        if s == "}":
            g.flush_phis()
            g.cur_block = None
            g.ind -= 1
            g.emit("}")
            continue
        if s.startswith("loop") or s.startswith("break") or s.startswith("continue"):
            raise ValueError("loops are not handled: " + s)
        m = re.match(r"@store_reg \((%\d+(?:\.[xyzw])?)(?: \([^)]*\))?, %(\d+)\)", s)
        if m:
            g.emit("r%s = %s;" % (m.group(2), g.val(m.group(1))))
            continue
        m = re.match(r"@store_deref \(%(\d+), (%\d+)\) \(wrmask=([xyzw]+)", s)
        if m:
            var = g.deref[m.group(1)]
            if var not in g.out_off:
                raise ValueError("store to " + var)
            src = m.group(2)[1:]
            t = g.types.get(src)
            for k, ch in enumerate(m.group(3)):
                comp = "xyzw".index(ch)
                e = "v%s[%d]" % (src, comp) if isinstance(t, tuple) else "s%s" % src
                g.emit("OUT[%d] = %s;" % (g.out_off[var] + comp, e))
            continue
        m = re.match(r"32(?:x(\d))?\s+%(\d+) = (.*)$", s)
        if not m:
            m1 = re.match(r"1\s+%(\d+) = (.*)$", s)
            if m1:
                m = re.match(r"32(?:x(\d))?\s+%(\d+) = (.*)$", "32 %" + m1.group(1) + " = " + m1.group(2))
            else:
                raise ValueError("line: " + s)
        ncomp, n, rhs = int(m.group(1) or 1), m.group(2), m.group(3)
        op = rhs.split(" ")[0]
        if op == "@decl_reg":
            g.types[n] = "reg"
            g.decl_v["r" + n] = 0
            continue
        if op == "undefined":
            # an output component the shader never writes (its store masks it out): zeros here, the caller's OUT keeps its own default
            if ncomp == 1:
                g.set(n, "0.0f")
            else:
                g.types[n] = ("v", ncomp)
                g.decl_v["v" + n] = ncomp
                for k in range(ncomp):
                    g.emit("v%s[%d] = 0.0f;" % (n, k))
            continue
        if op == "load_const":
            words = re.findall(r"0x([0-9a-f]{8})", rhs)
            g.const[n] = [int(w, 16) for w in words]
            if ncomp == 1:
                g.set(n, "RCN_BITS(0x%08xu)" % g.const[n][0])
            else:
                g.types[n] = ("v", ncomp)
                g.decl_v["v" + n] = ncomp
                for k in range(ncomp):
                    g.emit("v%s[%d] = RCN_BITS(0x%08xu);" % (n, k, g.const[n][k]))
            continue
        if op == "deref_var":
            g.deref[n] = re.match(r"deref_var &([\w#]+)", rhs).group(1)
            continue
        if op == "deref_array":   # MVPMatrix[i] in a vertex listing: only ever the address of the load_ubo that follows
            continue
        if op == "@load_deref":
            var = g.deref[re.match(r"@load_deref \(%(\d+)\)", rhs).group(1)]
            off = g.in_off[var]
            if ncomp == 1:
                g.set(n, "IN[%d]" % off)
            else:
                g.types[n] = ("v", ncomp)
                g.decl_v["v" + n] = ncomp
                for k in range(ncomp):
                    g.emit("v%s[%d] = IN[%d];" % (n, k, off + k))
            continue
        if op == "@load_ubo":
            mm = re.match(r"@load_ubo \(%\d+ \(0x0\), %\d+ \(0x([0-9a-f]+)\)\)", rhs)
            off = int(mm.group(1), 16) // 4
            if ncomp == 1:
                g.set(n, "U[%d]" % off)
            else:
                g.types[n] = ("v", ncomp)
                g.decl_v["v" + n] = ncomp
                for k in range(ncomp):
                    g.emit("v%s[%d] = U[%d];" % (n, k, off + k))
            continue
        if op == "@load_reg":
            g.set(n, "r" + re.match(r"@load_reg \(%(\d+)\)", rhs).group(1))
            continue
        if op == "phi":
            g.types[n] = "f"
            g.decl_v["PHI" + n] = 0
            g.set(n, "PHI" + n)
            continue
        if op.startswith("(float32)txf"):
            # texelFetchOffset: integer coordinates (a vec2 built from f2i32 values, kept as floats: exact below 2^24) + constant offset
            m2 = re.match(r"\(float32\)txf (%\d+) \(coord\), %\d+ \((0x[0-9a-f]+), (0x[0-9a-f]+)\) \(offset\), %\d+ \(0x0\) \(lod\), (\d+) \(texture\)", rhs)
            if not m2:
                raise ValueError("txf form: " + s)
            c = m2.group(1)[1:]
            g.types[n] = ("v", 4)
            g.decl_v["v" + n] = 4
            g.emit("RCN_TXF(TEXCTX, %s, (int)v%s[0] + %d, (int)v%s[1] + %d, v%s);" % (m2.group(4), c, int(m2.group(2), 16), c, int(m2.group(3), 16), n))
            continue
        if op.startswith("(float32)tex"):
            coord = re.match(r"\(float32\)tex (%\d+) \(coord\), (\d+) \(texture\)", rhs)
            c = coord.group(1)[1:]
            g.types[n] = ("v", 4)
            g.decl_v["v" + n] = 4
            g.emit("RCN_TEX(TEXCTX, %s, v%s[0], v%s[1], v%s);" % (coord.group(2), c, c, n))
            continue
        args = split_args(rhs[len(op):])
        if op in ("vec2", "vec3", "vec4"):
            g.types[n] = ("v", len(args))
            g.decl_v["v" + n] = len(args)
            for k, a in enumerate(args):
                g.emit("v%s[%d] = %s%s;" % (n, k, "(float)" if g.types.get(a.split(".")[0][1:]) == "i" else "", g.val(a)))
            continue
        if op in ALU1:
            g.set(n, ALU1[op].format(g.val(args[0])))
        elif op in ALU2:
            g.set(n, ALU2[op].format(g.val(args[0]), g.val(args[1])))
        elif op == "ffma":
            g.set(n, "RCN_FMA(%s, %s, %s)" % tuple(g.val(a) for a in args))
        elif op in CMP:
            g.set(n, CMP[op].format(g.val(args[0]), g.val(args[1])), "i")
        elif op in BOOL2:
            g.set(n, BOOL2[op].format(g.val(args[0]), g.val(args[1])), "i")
        elif op == "inot":
            g.set(n, "(!%s)" % g.val(args[0]), "i")
        elif op == "i2f32":
            # only met on integer uniforms (FrameCount), which the callers hand over as floats holding the integer's value
            src = args[0].split(".")[0][1:]
            if g.types.get(src) == "i":
                g.set(n, "(float)%s" % g.val(args[0]))
            else:
                g.set(n, g.val(args[0]))
        elif op == "f2i32":
            g.set(n, "RCN_F2I(%s)" % g.val(args[0]), "i")
        elif op == "b2f32":
            g.set(n, "(%s ? 1.0f : 0.0f)" % g.val(args[0]))
        elif op == "b32csel":
            ta = g.types.get(args[1].split(".")[0][1:])
            g.set(n, "(%s ? %s : %s)" % (g.val(args[0]), g.val(args[1]), g.val(args[2])), "i" if ta == "i" else "f")
        else:
            raise ValueError("unhandled op %s in: %s" % (op, s))
    out = []
    out.append("/* generated by oracle/glrun/nir2c.py from Mesa's NIR listing of the %s stage - do not edit */" % stage)
    out.append("#ifndef RCN_NO_TABLES")
    for kind, lst in (("uniforms", [(u[0], u[1], u[2], 0) for u in g.uniforms]),
                      ("inputs", [(i[0], g.in_off[i[0]], i[1], int(i[2])) for i in g.inputs]),
                      ("outputs", [(o[0], g.out_off[o[0]], o[1], int(o[2])) for o in g.outputs])):
        out.append("static const struct { const char* name; int off, n, flat; } %s_%s[] = {" % (name, kind))
        for nm, off, cnt, flat in lst:
            out.append('  {"%s", %d, %d, %d},' % (nm, off, cnt, flat))
        out.append("  {0, 0, 0, 0}};")
    out.append("#endif")
    out.append("#ifndef RCN_TABLES_ONLY")
    out.append("RCN_FN void %s(const float* U, const float* IN, float* OUT, void* TEXCTX) {" % name)
    fl = sorted(g.decl_f, key=int)
    for k in range(0, len(fl), 24):
        out.append("  float " + ", ".join("s" + x for x in fl[k:k + 24]) + ";")
    il = sorted(g.decl_i, key=int)
    for k in range(0, len(il), 24):
        out.append("  int " + ", ".join("b" + x for x in il[k:k + 24]) + ";")
    for v, cnt in sorted(g.decl_v.items()):
        out.append("  float %s%s;" % (v, "[%d]" % cnt if cnt else " = 0.0f"))
    out.append("  (void)U; (void)IN; (void)OUT; (void)TEXCTX;")
    out.extend(g.lines)
    out.append("}")
    out.append("#endif")
    return "\n".join(out) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("listing")
    ap.add_argument("--stage", choices=["vertex", "fragment"], required=True)
    ap.add_argument("--name", required=True)
    ap.add_argument("--index", type=int, default=-1, help="which shader of that stage in the listing (default: the last)")
    a = ap.parse_args()
    text = open(a.listing).read()
    tag = "shader: MESA_SHADER_" + a.stage.upper()
    parts = [m.start() for m in re.finditer(re.escape(tag), text)]
    if not parts:
        sys.exit("no %s shader in the listing" % a.stage)
    begin = parts[a.index]
    nxt = text.find("\nshader: ", begin + 1)
    sys.stdout.write(translate(text[begin:nxt if nxt > 0 else len(text)], a.stage, a.name))


if __name__ == "__main__":
    main()
