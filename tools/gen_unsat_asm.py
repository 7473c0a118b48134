#!/usr/bin/env python3
"""Generates ckb_zkp_amd/csrc/unsat_asm_gen.inc: the unsaturated-limb Montgomery product (unsat_dev.hpp) as ONE inline-asm
statement per product.

Why: hipcc lowers the C++ product scan to one chain per column that starts at 0 and adds the previous column's carry with a
separate 64-bit add (v_lshl_add_u64) — 17 extra VALU instructions per 9-limb product, 162 of the 2430 instructions of one
mixed addition in the G1 bucket-accumulation kernel — and it pads every inline-asm STATEMENT with an s_nop, so the fix cannot
be a pin per column.  Here the whole product is one statement: the carry of column k is the addend of column k+1's first
v_mad_u64_u32, the column accumulator lives in the clobbered pair v[0:1] (physical registers, so that its low word can feed
v_mul_lo_u32), the Montgomery factors m_i reuse the registers of the result limbs (m_i dies in column i + L - 1, r_i is born in
column i + L), and the last shift + move is one v_alignbit_b32.

    v_mad_u64_u32: 2 L^2 (+ L^2 per extra product of a lazily reduced sum) ; other VALU: 3 L + 2 (L - 1)
    L = 9: 162 + 43 = 205 instructions instead of 229 ; mul_add: 243 + 43

    python tools/gen_unsat_asm.py > ckb_zkp_amd/csrc/unsat_asm_gen.inc
"""
import sys

CFGS = [("Bn254Fq", 9, 29), ("Bls381Fq", 14, 28), ("Bn254Fr", 9, 29), ("Bls381Fr", 9, 29)]


def gen(L, B, nprod, square=False):
    """nprod products a_t * b_t summed, one reduction.  Operand numbering:
       %0..%(L-1)                 r_i   (out, early clobber; doubles as m_i)
       then for t < nprod:  a_t limbs (L), b_t limbs (L)     ["v"]
       then p limbs (L) ["s"], ninv ["s"]
    square (nprod == 1): b == a, cross terms once against a doubled copy held in extra early-clobber outputs d_i = 2 a_i."""
    mask = (1 << B) - 1
    lines = []
    r = lambda i: f"%{i}"
    base = L
    if square:
        d = lambda i: f"%{L + i}"          # doubled limbs (extra outputs)
        base = 2 * L
    a = lambda t, i: f"%{base + t * 2 * L + i}"
    b = lambda t, j: f"%{base + t * 2 * L + L + j}"
    pbase = base + nprod * 2 * L if not square else base + L
    if square:
        a = lambda t, i: f"%{base + i}"
    p = lambda j: f"%{pbase + j}"
    ninv = f"%{pbase + L}"
    if square:
        for i in range(L):
            lines.append(f"v_lshlrev_b32 {d(i)}, 1, {a(0, i)}")
    first = True
    for k in range(2 * L - 1):
        prods = []
        if square:
            for i in range(L):
                j = k - i
                if 0 <= j < L and i <= j:
                    prods.append((a(0, i), a(0, i)) if i == j else (d(i), a(0, j)))
        else:
            for t in range(nprod):
                for i in range(L):
                    j = k - i
                    if 0 <= j < L:
                        prods.append((a(t, i), b(t, j)))
        for i in range(L):
            j = k - i
            if 0 <= j < L and i < k:
                prods.append((r(i), p(j)))
        for (x, y) in prods:
            if first:
                lines.append(f"v_mad_u64_u32 v[0:1], vcc, {x}, {y}, 0")
                first = False
            else:
                lines.append(f"v_mad_u64_u32 v[0:1], vcc, {x}, {y}, v[0:1]")
        if k < L:
            lines.append(f"v_mul_lo_u32 {r(k)}, v0, {ninv}")
            lines.append(f"v_and_b32 {r(k)}, 0x{mask:x}, {r(k)}")
            lines.append(f"v_mad_u64_u32 v[0:1], vcc, {r(k)}, {p(0)}, v[0:1]")
            lines.append(f"v_lshrrev_b64 v[0:1], {B}, v[0:1]")
        else:
            lines.append(f"v_and_b32 {r(k - L)}, 0x{mask:x}, v0")
            if k < 2 * L - 2:
                lines.append(f"v_lshrrev_b64 v[0:1], {B}, v[0:1]")
            else:
                lines.append(f"v_alignbit_b32 {r(L - 1)}, v1, v0, {B}")
    return lines


def emit(name, pname, L, B, nprod, square=False):
    lines = gen(L, B, nprod, square)
    body = "\\n\\t".join(lines)
    outs = ", ".join(f'"=&v"(r.v[{i}])' for i in range(L))
    if square:
        outs += ", " + ", ".join(f'"=&v"(dbl[{i}])' for i in range(L))
        ins = ", ".join(f'"v"(a0.v[{i}])' for i in range(L))
        args = "const Fu<P>& a0"
    else:
        ins = ", ".join(", ".join(f'"v"(a{t}.v[{i}])' for i in range(L)) + ", " + ", ".join(f'"v"(b{t}.v[{i}])' for i in range(L))
                        for t in range(nprod))
        args = ", ".join(f"const Fu<P>& a{t}, const Fu<P>& b{t}" for t in range(nprod))
    ins += ", " + ", ".join(f'"s"(Fu<P>::mp_limb(1, {j}))' for j in range(L)) + ', "s"(Fu<P>::ninv())'
    nmad = sum(1 for l in lines if l.startswith("v_mad"))
    print(f"// {name}<{pname}>: {len(lines)} instructions, {nmad} v_mad_u64_u32")
    print(f"template <> struct {name}<{pname}> {{")
    print(f"  using P = {pname};")
    print(f"  static __device__ __forceinline__ Fu<P> run({args}) {{")
    print("    Fu<P> r;")
    if square:
        print(f"    uint32_t dbl[{L}];")
    print(f'    asm("{body}"')
    print(f"        : {outs}")
    print(f"        : {ins}")
    print('        : "v0", "v1", "vcc");')
    print("    return r;")
    print("  }")
    print("};")


def main():
    print("// GENERATED by tools/gen_unsat_asm.py — do not edit.  One inline-asm statement per unsaturated-limb Montgomery product:")
    print("// see the generator's docstring.  Included by unsat_dev.hpp inside namespace zkp when ZKP_UNSAT_ASM is defined.")
    print("template <class P> struct UnsatAsmMul;      // a0 * b0")
    print("template <class P> struct UnsatAsmMulAdd;   // a0 * b0 + a1 * b1, one reduction")
    print("template <class P> struct UnsatAsmSqr;      // a0 * a0, cross terms once")
    for pname, L, B in CFGS:
        emit("UnsatAsmMul", pname, L, B, 1)
        emit("UnsatAsmMulAdd", pname, L, B, 2)
        emit("UnsatAsmSqr", pname, L, B, 1, square=True)


if __name__ == "__main__":
    main()
