#!/usr/bin/env python3
"""Opcode histogram of the gfx950 kernels in libbbs_sign_amd.so (static, from the ISA), the input of tools/valu_model.py.

    tools/isa_histogram.py bbs_sign_amd/libbbs_sign_amd.so profiles/r02_isa_histogram.csv [regex ...]

For every selected kernel (default: the BLS12-381 PairDist and PvMsmPart kernels and the device functions they call:
d_mul, d_inv, d_final_exp, d_frob, d_pow_xabs_to ...) it writes two mixes:
  scope=text  : every instruction of the symbol
  scope=loops : only instructions inside a loop of the symbol (between a backward branch and its target) -- the hot
                code; straight-line set-up code is executed once per wavefront, loop bodies 63 .. 10^3 times.
The mix used by the issue model is scope=loops of the kernel plus scope=text of its callees (they are called from
inside the kernel's loops).  Columns: kernel,symbol,scope,opcode,count.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def extract_code_objects(lib, tmp):
    subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + tmp + "/fat.bin", lib], check=True)
    data = open(tmp + "/fat.bin", "rb").read()
    idx = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    out = []
    for n, i in enumerate(idx):
        b = tmp + "/b%d.bin" % n
        open(b, "wb").write(data[i:(idx[n + 1] if n + 1 < len(idx) else len(data))])
        co = b + ".co"
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + b, "--output=" + co, "--unbundle"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def disassemble(co):
    """-> {symbol: [(addr, opcode, operands)]}"""
    txt = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], stdout=subprocess.PIPE, text=True, check=True).stdout
    syms, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            cur = m.group(2)
            syms[cur] = []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            syms[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return syms


def loop_mask(insts):
    """True for instructions inside a backward-branch loop of this symbol (target = pc + 4 + 4 * simm16)."""
    addr_index = {a: i for i, (a, _, _) in enumerate(insts)}
    inside = [False] * len(insts)
    for i, (a, op, args) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            try:
                imm = int(args.split()[0])
            except (ValueError, IndexError):
                continue
            if imm >= 0x8000:
                imm -= 0x10000
            target = a + 4 + 4 * imm
            if target <= a and target in addr_index:
                for k in range(addr_index[target], i + 1):
                    inside[k] = True
    return inside


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True)
    return dict(zip(names, r.stdout.splitlines()))


def main():
    lib, out = sys.argv[1], sys.argv[2]
    pats = sys.argv[3:] or [r"k_stage<bbs::PairDist<bbs::BlsCurve>", r"k_stage<bbs::PvMsmPart<bbs::BlsCurve>", r"k_stage<bbs::PvChallenge<bbs::BlsCurve>",
                            r"k_stage<bbs::PvScalars<bbs::BlsCurve>", r"k_stage<bbs::PvFinish,", r"k_stage<bbs::PairMillerHalf<bbs::BlsCurve>",
                            r"k_stage<bbs::PairFinalDist<bbs::BlsCurve>", r"k_stage<bbs::PvIngest<bbs::BlsCurve>"]
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in extract_code_objects(os.path.abspath(lib), tmp):
            syms = disassemble(co)
            dm = demangle(list(syms))
            kernels = [s for s in syms if any(re.search(p, dm.get(s, s)) for p in pats)]
            if not kernels:
                continue
            for k in kernels:
                kname = re.sub(r"void rt::k_stage<bbs::(\w+)<bbs::(\w+)>.*", r"\1<\2>", dm[k])
                kname = re.sub(r"void rt::k_stage<bbs::(\w+),.*", r"\1", kname)          # non-template stages (PvFinish)
                curve = "BlsCurve" if "BlsCurve" in dm[k] else "BnCurve"
                # the device functions a kernel can reach, by name: the lane-sliced Fp12 routines for the pairing kernels, the
                # hash routines only for the stages that hash, everything else of the curve (field / group arithmetic) otherwise
                pair_fn = lambda d: any(t in d for t in ("d_mul", "d_inv", "d_final", "d_frob", "d_pow", "d_gather"))
                hash_fn = lambda d: "sha256" in d or "xmd" in d
                hashes = any(t in kname for t in ("PvChallenge", "PvScalars"))
                leaf = any(t in kname for t in ("PvFinish", "PvIngest", "PairMillerHalf"))
                group = [k] + [s for s in syms if s != k and not leaf and not dm.get(s, s).startswith("void rt::k_stage")
                               and (curve in dm.get(s, s) or (hashes and hash_fn(dm.get(s, s))))
                               and ("Pair" in kname) == pair_fn(dm.get(s, s)) and (hashes or not hash_fn(dm.get(s, s)))]
                for s in group:
                    insts = syms[s]
                    if not insts:
                        continue
                    inside = loop_mask(insts)
                    sname = re.sub(r"\(.*", "", dm.get(s, s))[:80]
                    for scope, sel in (("text", [True] * len(insts)), ("loops", inside)):
                        cnt = collections.Counter(op for (a, op, _), f in zip(insts, sel) if f)
                        for op, c in sorted(cnt.items(), key=lambda x: -x[1]):
                            rows.append((kname, sname, scope, op, c))
    with open(out, "w") as f:
        f.write("# static opcode counts from llvm-objdump -d of the gfx950 code objects in %s (tools/isa_histogram.py)\n" % os.path.basename(lib))
        f.write("kernel,symbol,scope,opcode,count\n")
        for r in rows:
            f.write('%s,"%s",%s,%s,%d\n' % r)
    print("wrote", out, len(rows), "rows")


if __name__ == "__main__":
    main()
