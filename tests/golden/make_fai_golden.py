#!/usr/bin/env python3
"""Writes tests/golden/fai_cases/*.fa (crafted FASTA files) and, beside each and beside every FASTA under
tests/golden/inputs, the .fai that the REAL `cora faiGenerate` command produces for it.

The command is not restated here: it is taken at run time from the string literal in
/root/reference/cora.cpp (the awk one-liner of faiGenerate, cora.cpp:1023) and run with this image's awk, as
cora itself would run it.  Needs /root/reference; run in the build container only.  The outputs are data
(expected .fai text) and are committed.
"""
import os, re, subprocess, random

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = os.path.join(HERE, "fai_cases")


def reference_command():
    src = open("/root/reference/cora.cpp", encoding="latin-1").read()
    m = re.search(r'faiGenerateCall\s*<<\s*"((?:[^"\\]|\\.)*)"\s*;', src)
    assert m, "faiGenerate command not found in cora.cpp"
    lit = m.group(1)
    out, i = [], 0
    while i < len(lit):                       # C string literal -> text
        if lit[i] == "\\":
            out.append({"n": "\n", "t": "\t", '"': '"', "\\": "\\"}[lit[i + 1]])
            i += 2
        else:
            out.append(lit[i])
            i += 1
    return "".join(out)


def crafted():
    rnd = random.Random(7)
    seq = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
    wrap = lambda s, w: "".join(s[i:i + w] + "\n" for i in range(0, len(s), w))
    c = {}
    c["plain60"] = ">chr1\n" + wrap(seq(200), 60) + ">chr2\n" + wrap(seq(120), 60)
    c["descr_tab"] = ">chr1 some description\there\n" + wrap(seq(130), 50) + ">chr2\tx\n" + wrap(seq(50), 50)
    c["no_final_newline"] = (">a\n" + wrap(seq(95), 40) + ">b\n" + wrap(seq(41), 40))[:-1]
    c["ragged"] = ">r1\n" + "".join(seq(n) + "\n" for n in (10, 33, 1, 60, 7)) + ">r2\n" + seq(300) + "\n"
    c["blank_lines"] = ">x\n" + seq(20) + "\n\n" + seq(20) + "\n   \n" + ">y\n\n" + seq(20) + "\n\n"
    c["leading_blanks"] = "  >lead\n \t" + seq(30) + "  trailing words\n" + seq(30) + " \n>z\n" + seq(5) + "\n"
    c["crlf"] = ">c1\r\n" + "".join(seq(25) + "\r\n" for _ in range(3)) + ">c2\r\n" + seq(9) + "\r\n"
    c["consecutive_headers"] = ">e1\n>e2\n" + wrap(seq(70), 30) + ">e3\n>e4\n"
    c["text_before_header"] = seq(17) + "\n" + seq(5) + "\n>late\n" + wrap(seq(64), 16)
    c["no_header"] = wrap(seq(100), 25)
    c["empty"] = ""
    c["only_newlines"] = "\n\n\n"
    c["header_only"] = ">solo"
    c["gt_inside"] = ">g\nAC>GT\n>h extra\nA\n"
    c["lowercase_iupac"] = ">m\n" + wrap("acgtnNRYKMacgt" * 9, 21)
    c["long_single_line"] = ">one\n" + seq(5000) + "\n>two\n" + seq(3) + "\n"
    many = []
    for i in range(300):
        many.append(">ctg%d\n" % i + wrap(seq(rnd.randint(1, 150)), 70))
    c["many_contigs"] = "".join(many)
    big = []
    for i in range(5):                         # crosses several 4096-byte scan tiles, ragged
        big.append(">big%d d\n" % i)
        for _ in range(rnd.randint(150, 400)):
            big.append(seq(rnd.choice((60, 60, 60, 59, 61, 0, 120))) + rnd.choice(("\n", "\n", " \n", "\t#\n")))
    c["multi_tile_ragged"] = "".join(big)
    return c


def main():
    cmd = reference_command()
    os.makedirs(CASES, exist_ok=True)
    for name, text in crafted().items():
        with open(os.path.join(CASES, name + ".fa"), "wb") as f:
            f.write(text.encode("latin-1"))
    targets = [os.path.join(CASES, n) for n in sorted(os.listdir(CASES)) if n.endswith(".fa")]
    inputs = os.path.join(HERE, "inputs")
    outdir = os.path.join(HERE, "fai_inputs")
    os.makedirs(outdir, exist_ok=True)
    for n in sorted(os.listdir(inputs)):
        if n.endswith(".fa"):
            targets.append(os.path.join(inputs, n))
    for fa in targets:
        dst = fa + ".fai" if fa.startswith(CASES) else os.path.join(outdir, os.path.basename(fa) + ".fai")
        subprocess.check_call(cmd + fa + " > " + dst, shell=True)
        print(os.path.relpath(dst, HERE), os.path.getsize(dst))


if __name__ == "__main__":
    main()
