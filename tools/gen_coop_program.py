#!/usr/bin/env python3
"""Generates plutus_halo2_verifier_gen_amd/csrc/coop_program.h: the fixed sequence of Fp12-level operations the
cooperative pairing kernel interprets (Miller loop over precomputed lines + final exponentiation), after checking
it by simulation against the package's big-integer pairing.

The program is the same for every plan: which line tables / points it touches comes from the kernel arguments.
  accept <=> e(P1, Q1) == e(P2', Q2)  evaluated as FE( ML(P1; lines(Q1)) * ML(-P2'; lines(Q2)) ) == 1
  final exponentiation: easy part, then 3(p^4-p^2+1)/r = (x-1)^2 (x+p)(x^2+p^2-1) + 3.
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plutus_halo2_verifier_gen_amd import bls12_381 as bls  # noqa: E402

# opcodes (4 bytes per instruction: op, dst, a, b)
OP_END, OP_MUL, OP_MSTEP, OP_CONJ, OP_FROB, OP_INV, OP_MOV, OP_SETONE, OP_DUMP, OP_WARMUP, OP_CSQR, OP_MILLER, OP_EXPX = range(13)
OP_NAMES = ["END", "MUL", "MSTEP", "CONJ", "FROB", "INV", "MOV", "SETONE", "DUMP", "WARMUP", "CSQR", "MILLER", "EXPX"]
# MILLER: the whole Miller loop on F as ONE interpreter step (63 x MSTEP with the line counts read off the bits of |x|): F
# stays in registers from the first squaring to the last line instead of going through the variable file 63 times.
# EXPX d, a: d = conj(a^|x|) = a^x for a in the cyclotomic subgroup, as one step: the running power stays in registers
# through the runs of cyclotomic squarings and the five multiplications by a (re-read from the variable file each time).
# MSTEP n, line: one step of the Miller loop on F: F = F^2 (general squaring), then n in {1, 2} times { F *= line of
# loop 1, F *= line of loop 2 } starting at line index `line` (doubling step; plus the addition step where the bit of
# |x| is set).  F stays in registers for the whole step.
# CSQR d, a, n: d = a^(2^n) by n cyclotomic squarings; the value stays in registers between them (a run of squarings
# is one interpreter step: no round trip of the variable through private memory per squaring)
# variables
# (five variables: the running power S is overwritten step by step and the result is built in F - every variable is a slice of
# each lane's private memory, which sizes the queue's scratch arena)
F, A, T, U, S = range(5)
T0 = T1 = T2 = S
T3 = F
N_VARS = 5


def build_program():
    prog = []
    e = lambda *ins: prog.append(tuple(list(ins) + [0] * (4 - len(ins))))
    e(OP_SETONE, F)
    e(OP_WARMUP)
    e(OP_MILLER)
    assert sum(2 if bit else 1 for bit in bls.miller_bits()) == 68
    # x < 0: the Miller value is m = conj(F).  Easy part m^(p^6-1) = conj(m) * m^-1 = F * conj(F^-1); written so that
    # every CONJ / INV operand is an engine output (the lazily reduced field bounds of h2v_pairing_coop.hpp).
    e(OP_CONJ, U, F)
    e(OP_DUMP, 0, U)
    e(OP_INV, A, F)
    e(OP_CONJ, A, A)
    e(OP_MUL, T, F, A)        # m^(p^6-1)
    e(OP_FROB, A, T)
    e(OP_FROB, A, A)
    e(OP_MUL, T, A, T)        # ^(p^2+1)

    def exp_x(dst, src):      # dst = src^x, x = -|x|  (cyclotomic subgroup: inverse = conjugate)
        e(OP_EXPX, dst, src)

    exp_x(A, T); e(OP_CONJ, U, T); e(OP_MUL, T0, A, U)       # t^(x-1)
    exp_x(A, T0); e(OP_CONJ, U, T0); e(OP_MUL, T1, A, U)     # ^(x-1)
    exp_x(A, T1); e(OP_FROB, U, T1); e(OP_MUL, T2, A, U)     # ^(x+p)
    exp_x(A, T2); e(OP_MOV, U, A); exp_x(A, U)               # t2^(x^2)
    e(OP_FROB, U, T2); e(OP_FROB, U, U); e(OP_MUL, T3, A, U)
    e(OP_CONJ, U, T2); e(OP_MUL, T3, T3, U)                  # ^(x^2+p^2-1)
    e(OP_MUL, U, T, T); e(OP_MUL, U, U, T)                   # t^3
    e(OP_MUL, T3, T3, U)
    e(OP_DUMP, 1, T3)         # (T3 is F: the result is read from F)
    e(OP_END)
    return prog


def check_bounds(prog):
    """Static check of the value bounds the kernel's lazily reduced field relies on (h2v_pairing_coop.hpp): with
    `v` = the multiple of p a variable may reach, engine results are 3, CONJ needs v <= 5 and gives 6, FROB gives 5,
    and every staged operand needs v <= 6."""
    v = [None] * N_VARS
    for op, d, a, b in prog:
        if op == OP_END:
            break
        if op == OP_SETONE:
            v[d] = 1
        elif op == OP_MUL:
            assert v[a] <= 6 and v[b] <= 6
            v[d] = 3
        elif op == OP_CSQR:
            assert v[a] <= 6 and b >= 1
            v[d] = 3
        elif op == OP_MSTEP:
            assert v[F] <= 6 and d in (1, 2)
            v[F] = 3
        elif op == OP_MILLER:
            assert v[F] <= 6
            v[F] = 3
        elif op == OP_EXPX:
            assert v[a] <= 6          # staged as an operand of the squarings and of the multiplications
            v[d] = 6                  # conj of an engine result
        elif op == OP_CONJ:
            assert v[a] <= 5, "CONJ of a value that is not an engine / FROB result"
            v[d] = 6
        elif op == OP_FROB:
            assert v[a] <= 6
            v[d] = 5
        elif op == OP_INV:
            assert v[a] <= 5
            v[d] = 3
        elif op == OP_MOV:
            v[d] = v[a]
        elif op == OP_WARMUP:
            assert v[F] <= 6
        elif op == OP_DUMP:
            assert v[a] <= 1024
    return True


def simulate(prog, p1, q1, p2, q2):
    """Runs the program with big integers.  p2 is used negated (the kernel is handed -er).  Returns var F."""
    lines = {1: bls.g2_line_table(q1) if p1 is not None else None, 2: bls.g2_line_table(q2) if p2 is not None else None}
    pts = {1: p1, 2: bls.g1_neg(p2)}
    v = [None] * N_VARS
    for op, d, a, b in prog:
        if op == OP_END:
            break
        if op == OP_SETONE:
            v[d] = list(bls.F12_ONE)
        elif op == OP_WARMUP or op == OP_DUMP:
            pass
        elif op == OP_MUL:
            v[d] = bls.f12_mul(v[a], v[b])
        elif op == OP_CSQR:
            t = v[a]
            for _ in range(b):
                t = bls.f12_sqr(t)
            v[d] = t
        elif op == OP_MSTEP:
            v[F] = bls.f12_sqr(v[F])
            for idx in range(a, a + d):
                for loop in (1, 2):
                    if pts[loop] is not None:
                        v[F] = bls.f12_mul(v[F], bls._line_eval(lines[loop][idx], pts[loop]))
        elif op == OP_MILLER:
            idx = 0
            for bit in bls.miller_bits():
                v[F] = bls.f12_sqr(v[F])
                for _ in range(2 if bit else 1):
                    for loop in (1, 2):
                        if pts[loop] is not None:
                            v[F] = bls.f12_mul(v[F], bls._line_eval(lines[loop][idx], pts[loop]))
                    idx += 1
        elif op == OP_EXPX:
            t = v[a]
            for bit in bls.miller_bits():
                t = bls.f12_sqr(t)
                if bit:
                    t = bls.f12_mul(t, v[a])
            v[d] = bls.f12_conj(t)
        elif op == OP_CONJ:
            v[d] = bls.f12_conj(v[a])
        elif op == OP_FROB:
            v[d] = bls.f12_frob(v[a])
        elif op == OP_INV:
            v[d] = bls.f12_inv(v[a])
        elif op == OP_MOV:
            v[d] = list(v[a])
        else:
            raise ValueError(op)
    return v[F]


def self_check():
    prog = build_program()
    assert check_bounds(prog)
    rng = random.Random(3)
    s = rng.randrange(2, bls.R)
    q1 = bls.g2_mul(bls.G2_GEN, s)
    for trial in range(3):
        a = rng.randrange(1, bls.R)
        pa = bls.g1_mul(bls.G1_GEN, a)
        spa = bls.g1_mul(pa, s)
        bad = bls.g1_add(spa, bls.G1_GEN)
        assert simulate(prog, pa, q1, spa, bls.G2_GEN) == bls.F12_ONE
        assert simulate(prog, pa, q1, bad, bls.G2_GEN) != bls.F12_ONE
    assert simulate(prog, None, q1, None, bls.G2_GEN) == bls.F12_ONE
    assert simulate(prog, bls.G1_GEN, q1, None, bls.G2_GEN) != bls.F12_ONE
    # the program's result is the canonical final exponentiation cubed
    pa = bls.g1_mul(bls.G1_GEN, 5)
    f = bls.f12_mul(bls.miller_loop(pa, q1), bls.miller_loop(bls.g1_neg(bls.G1_GEN), bls.G2_GEN))
    fe = bls.final_exponentiation(f)
    assert simulate(prog, pa, q1, bls.G1_GEN, bls.G2_GEN) == bls.f12_mul(bls.f12_mul(fe, fe), fe)
    return prog


def emit():
    prog = self_check()
    o = ["// GENERATED by tools/gen_coop_program.py (program simulated against the big-integer pairing) - do not edit.",
         "#pragma once", "#include <stdint.h>"]
    for i, n in enumerate(OP_NAMES):
        o.append("#define COOP_OP_%s %d" % (n, i))
    o.append("#define COOP_N_VARS %d" % N_VARS)
    o.append("#define COOP_VAR_F %d" % F)
    o.append("#define COOP_PROGRAM_LEN %d" % len(prog))
    o.append("__device__ const uint32_t COOP_PROGRAM[%d] = {" % len(prog))
    row = []
    for op, d, a, b in prog:
        row.append("0x%08xu" % (op | (d << 8) | (a << 16) | (b << 24)))
        if len(row) == 8:
            o.append("    " + ", ".join(row) + ",")
            row = []
    if row:
        o.append("    " + ", ".join(row) + ",")
    o.append("};")
    bits = bls.miller_bits()
    n_expx = sum(1 for p in prog if p[0] == OP_EXPX)
    n_mil = sum(1 for p in prog if p[0] == OP_MILLER)
    n_mul = sum(1 for p in prog if p[0] == OP_MUL) + n_expx * sum(1 for b in bits if b)
    n_line = sum(2 * p[1] for p in prog if p[0] == OP_MSTEP) + n_mil * 2 * sum(2 if b else 1 for b in bits)
    n_csqr = sum(p[3] for p in prog if p[0] == OP_CSQR) + n_expx * len(bits)
    n_sqr = sum(1 for p in prog if p[0] == OP_MSTEP) + n_mil * len(bits)
    o.append("// %d instructions: %d MUL, %d SQR, %d CSQR, %d LINE" % (len(prog), n_mul, n_sqr, n_csqr, n_line))
    path = os.path.join(ROOT, "plutus_halo2_verifier_gen_amd", "csrc", "coop_program.h")
    with open(path, "w") as f:
        f.write("\n".join(o) + "\n")
    print("wrote", path, len(prog), "instructions,", n_mul, "MUL,", n_line, "LINE")


if __name__ == "__main__":
    emit()
