#!/usr/bin/env python3
"""Loop census of one kernel in a gfx950 assembly listing (hipcc --cuda-device-only -S): for every backward branch,
the instruction mix of the range [target label, branch].  Usage: isa_loops.py engine.s '<demangled substring>'"""
import collections
import re
import subprocess
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if "@function" in l and ".type" in l:
            name = l.split()[1].rstrip(",").split(",")[0]
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            if want in dem:
                start = i
                print("kernel:", dem[:200])
                break
    if start is None:
        raise SystemExit("not found")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    body = lines[start:end]
    labels, instrs = {}, []
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"(\.LBB\d+_\d+):", t)
            if m:
                labels[m.group(1)] = len(instrs)
            continue
        instrs.append(t.split(";")[0].strip())
    print("instructions:", len(instrs))
    loops = []
    for k, ins in enumerate(instrs):
        m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", ins)
        if m and m.group(1) in labels and labels[m.group(1)] <= k:
            loops.append((labels[m.group(1)], k, m.group(1)))
    for a, b, lab in sorted(loops):
        mix = collections.Counter()
        for ins in instrs[a:b + 1]:
            op = ins.split()[0]
            if op.startswith("v_pk_"):
                mix["v_pk"] += 1
            elif op.startswith("v_cvt"):
                mix["v_cvt"] += 1
            elif op.startswith("v_cndmask") or op.startswith("v_cmp"):
                mix["v_cmp/cnd"] += 1
            elif op.startswith(("v_fma", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_mac", "v_fmac")):
                mix["v_f32"] += 1
            elif op.startswith(("v_rsq", "v_rcp", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")):
                mix["v_trans"] += 1
            elif op.startswith("v_mov") or op.startswith("v_accvgpr"):
                mix["v_mov"] += 1
            elif op.startswith("v_"):
                mix["v_int/other"] += 1
            elif op.startswith("ds_"):
                mix["ds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                mix[op.split("_")[0]] += 1
            elif op.startswith("s_waitcnt"):
                mix["s_wait"] += 1
            elif op.startswith("s_"):
                mix["s_other"] += 1
            else:
                mix["other"] += 1
        valu = sum(v for k2, v in mix.items() if k2.startswith("v_"))
        print("loop %s: instr %d..%d (%d), VALU %d: %s" % (lab, a, b, b - a + 1, valu, dict(mix)))


if __name__ == "__main__":
    main()
