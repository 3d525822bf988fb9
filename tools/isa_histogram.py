"""Histogram of one kernel's instructions in a hipcc `-save-temps` .s file.

usage: python tools/isa_histogram.py file.s kernel_name [--top N] [--dump]
Groups by opcode, and totals VALU / SALU / VMEM / LDS counts plus a VALU *issue-slot* estimate
(transcendentals and 32-bit integer multiplies are quarter rate on CDNA: 4 slots each).
Static counts: straight-line kernels (the sweeps) execute every instruction once per wave
except inside the branches/loops visible with --dump.
"""
import collections
import re
import sys

QUARTER = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_",
           "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32")


def kernel_lines(path, name):
    out, on = [], False
    for line in open(path):
        if line.startswith(name + ":"):
            on = True
            continue
        if on:
            if line.startswith(".Lfunc_end") or line.strip().startswith(".end_amdhsa_kernel"):
                break
            out.append(line.rstrip("\n"))
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    lines = kernel_lines(path, name)
    if "--dump" in sys.argv:
        print("\n".join(lines))
        return
    ops = collections.Counter()
    for l in lines:
        m = re.match(r"\s+([a-z_0-9]+)\s", l + " ")
        if m and not l.strip().startswith((";", ".")):
            ops[m.group(1)] += 1
    cls = collections.Counter()
    slots = 0
    for op, c in ops.items():
        if op.startswith("v_"):
            cls["VALU"] += c
            slots += c * (4 if op.startswith(QUARTER) else 1)
        elif op.startswith("s_"):
            cls["SALU/ctl"] += c
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cls["VMEM"] += c
        elif op.startswith("ds_"):
            cls["LDS"] += c
        else:
            cls["other"] += c
    print(f"{name}: {sum(ops.values())} instructions  " + "  ".join(f"{k}={v}" for k, v in cls.items())
          + f"  VALU issue slots~{slots}")
    for op, c in ops.most_common(top):
        print(f"  {c:5d}  {op}")


if __name__ == "__main__":
    main()
