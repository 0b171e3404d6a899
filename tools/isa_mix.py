"""Static instruction mix of the kernels in one csrc/*.hip file, weighted with the gfx950 issue costs measured by
tools/micro/valu_bench.hip (profiles/r02_valu_issue_bench.txt): 4 SIMD cycles for v_mad_u64_u32, 64-bit adds / compares,
carry arithmetic, 32-bit multiplies and three-operand VOP3, 2 for the other 32-bit vector instructions.

    python tools/isa_mix.py ntt.hip [kernel-substring]
    python tools/isa_mix.py --csv out.csv file.hip[:substring] ...      kernel, VALU instructions, issue cycles, cycles per instruction

Straight-line kernels only (loops are counted once), so use it for ratios: how much of a kernel's issue time is the
multiply-adds, how much the glue around them.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "certificate-stark_amd"))
import build as B  # noqa: E402

FOUR = re.compile(r"^v_(mad_u64_u32|mad_i64_i32|lshl_add_u64|cmp_\w+_[ui]64|cmpx_\w+_[ui]64|add_co_u32|addc_co_u32|sub_co_u32|subb_co_u32|"
                  r"subrev_co_u32|subbrev_co_u32|mul_lo_u32|mul_hi_u32|mul_hi_i32|add3_u32|alignbit_b32|lshlrev_b64|lshrrev_b64|ashrrev_i64|"
                  r"perm_b32|bfe_u32|bfi_b32|and_or_b32|or3_b32|xad_u32|lshl_or_b32|lshl_add_u32|add_lshl_u32|mad_u32_u24|mad_i32_i24)")


def classify(op):
    if op.startswith("v_mad_u64_u32"):
        return "mad64", 4
    if op.startswith(("v_lshl_add_u64", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_subrev_co", "v_subbrev_co")):
        return "add64/carry", 4
    if re.match(r"^v_cmpx?_\w+_[ui]64", op):
        return "cmp64", 4
    if op.startswith(("v_mov_b32", "v_accvgpr")):
        return "mov", 2
    if op.startswith("v_cndmask"):
        return "select", 2
    if FOUR.match(op):
        return "other4", 4
    if op.startswith("v_"):
        return "other2", 2
    if op.startswith(("ds_", "global_", "buffer_", "scratch_", "flat_")):
        return "mem", 0
    return "scalar", 0


def main():
    if sys.argv[1] == "--csv":
        rows, wants = [], collections.OrderedDict()
        for spec in sys.argv[3:]:  # file.hip:substring -- every file is compiled once
            src, _, want = spec.partition(":")
            wants.setdefault(src, []).append(want)
        for src, ws in wants.items():
            rows += one(src, ws, quiet=True)
        with open(sys.argv[2], "w") as f:
            f.write("kernel,valu_instructions_static,issue_cycles_static,cycles_per_valu_instruction\n")
            for name, valu, total in rows:
                f.write('"%s",%d,%d,%.3f\n' % (name, valu, total, total / max(valu, 1)))
        print(open(sys.argv[2]).read())
        return
    one(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")


def one(src, want, quiet=False):
    rows = []
    path = src if os.path.exists(src) else os.path.join(B.CSRC, src)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        flags = [f for f in B.FLAGS if f != "-fPIC"]
        subprocess.check_call([B.hipcc()] + flags + ["-S", "--cuda-device-only", "-o", out, path], stderr=subprocess.DEVNULL)
        text = open(out).read()
    kernels = re.findall(r"^(_Z\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M)
    for name, body in kernels:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if want and not any(w in dem for w in ([want] if isinstance(want, str) else want)):
            continue
        counts, cycles = collections.Counter(), collections.Counter()
        for line in body.splitlines():
            line = line.strip()
            if not line or line.startswith((";", ".")) or line.endswith(":"):
                continue
            op = line.split()[0]
            cls, w = classify(op)
            counts[cls] += 1
            cycles[cls] += w
        total = sum(cycles.values())
        valu = sum(c for k, c in counts.items() if k not in ("mem", "scalar"))
        short = dem.replace("cs::(anonymous namespace)::", "").replace("void ", "").replace("cs::", "")
        short = re.sub(r"\(.*$", "", short).strip()  # the name as the profiler's per-kernel summaries (tools/pmc_summary.py) spell it
        rows.append((short, valu, total))
        if not quiet:
            print("%-44s valu %6d  cycles %7d  " % (short[:44], valu, total) +
              "  ".join("%s %d%%" % (k, round(100.0 * cycles[k] / max(total, 1))) for k in
                        ("mad64", "add64/carry", "cmp64", "mov", "select", "other4", "other2")) + "  mem %d" % counts["mem"])
    return rows


if __name__ == "__main__":
    main()
