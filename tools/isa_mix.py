"""VALU opcode mix of a kernel of librt_hip (hipcc -S): how many of its vector instructions belong to the class the gfx950
SIMD issues in ~2 cycles per wave64 (v_fma/mul/add/sub_f32, v_mov, v_and/or/xor, v_add/sub_u32, shifts: measured by
tools/issue_microbench.hip) and how many to the ~4-cycle class (v_min/max/min3/max3, compares, v_cndmask, packed fp32, fp64,
64-bit adds, conversions).

  python tools/isa_mix.py <mangled-name substring> [...]                     static mix of the whole kernel
  python tools/isa_mix.py --loops <substring> [--weights C=29,B=10.7,A=4.8]  per LOOP of a three-loop trace kernel, and the
        mix weighted by how often each loop's body runs (passes per 64 rays from tools/wave_schedule_model.py): the DYNAMIC mix
        tools/make_counters_json.py prices the vector ALU's ceiling with (the static mix rounded to 0.5 before).
        Loops are recognised by content: C = the depth-2 loop that converts bytes (v_cvt_f32_ubyte: the wide-node visit),
        B = the depth-2 loop that divides (v_div_scale / v_rcp: the ray-triangle test), A = the rest of the outer loop.
        Every instruction of a loop body counts once per pass (branches over pushes / pops are not modelled)."""
import os, re, subprocess, sys, tempfile, collections, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = re.compile(r"^v_(fma_f32|fmac_f32|fmamk_f32|fmaak_f32|mul_f32|add_f32|sub_f32|subrev_f32|mov_b32|and_b32|or_b32|xor_b32|add_u32|sub_u32|subrev_u32|"
                  r"lshlrev_b32|lshrrev_b32|ashrrev_i32|and_or_b32|lshl_or_b32|lshl_add_u32|add_lshl_u32|or3_b32|bfe_u32|add3_u32)")


def disassemble():
    out = os.path.join(tempfile.mkdtemp(prefix="isa_"), "rt_hip.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "raytracing_amd", "csrc"), "--cuda-device-only", "-S",
                           "-o", out, os.path.join(ROOT, "raytracing_amd", "csrc", "rt_hip.hip")], stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernel_lines(text, want):
    inside, got = False, []
    for line in text:
        if re.match(r"^_Z\w*" + re.escape(want) + r"\w*:", line):
            inside = True
            continue
        if inside:
            if line.startswith(".Lfunc_end"):
                break
            got.append(line)
    return got


def op_of(line):
    m = re.match(r"^\s+(v_[a-z0-9_]+)", line)
    return m.group(1).replace("_e32", "").replace("_e64", "") if m else None


def static_mix(lines):
    ops = collections.Counter(o for o in map(op_of, lines) if o)
    total = sum(ops.values())
    fast = sum(n for o, n in ops.items() if FAST.match(o))
    return total, fast, ops


def loop_mix(lines):
    """{loop name: (total VALU, fast VALU)} for A / B / C (see the module docstring)"""
    parent2 = {}                    # header of a deeper loop -> its depth-2 ancestor
    cur = None                      # depth-2 loop (header label) the current block belongs to, or None (outer loop / outside)
    per = collections.defaultdict(lambda: [0, 0, collections.Counter()])
    pending_label, parents = None, {}
    for line in lines:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", line)
        if m:
            label, comment = m.group(1), m.group(2) or ""
            pending_label, parents = label, {}
            mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", comment)
            if mm:
                h, d = "." + "L" + mm.group(1), int(mm.group(2))
                cur = None if d < 2 else (h if d == 2 else parent2.get(h))
            pm = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", comment)
            if pm:
                parents[int(pm.group(2))] = ".L" + pm.group(1)
            continue
        pm = re.search(r";\s+Parent Loop (BB\d+_\d+) Depth=(\d+)", line)
        if pm and pending_label:
            parents[int(pm.group(2))] = ".L" + pm.group(1)
            continue
        hm = re.search(r"This (Inner )?Loop Header: Depth=(\d+)", line)
        if hm and pending_label:
            d = int(hm.group(2))
            if d == 2:
                cur = pending_label
            elif d > 2:
                parent2[pending_label] = parents.get(2)
                cur = parents.get(2)
            else:
                cur = None
            continue
        o = op_of(line)
        if o:
            e = per[cur]
            e[0] += 1
            e[1] += 1 if FAST.match(o) else 0
            e[2][o] += 1
    named = {}
    for k, (tot, fast, ops) in per.items():
        if k is None:
            name = "A"
        elif any(o.startswith("v_cvt_f32_ubyte") for o in ops):
            name = "C"
        elif any(o.startswith("v_div_scale") or o.startswith("v_rcp") or o.startswith("v_div_fixup") for o in ops):
            name = "B"
        else:
            name = "A"
        t = named.setdefault(name, [0, 0])
        t[0] += tot
        t[1] += fast
    return named


if __name__ == "__main__":
    argv = sys.argv[1:]
    text = disassemble()
    if argv and argv[0] == "--loops":
        weights = dict(C=29.0, B=10.7, A=4.8)
        names = []
        i = 1
        while i < len(argv):
            if argv[i] == "--weights":
                weights = {k: float(v) for k, v in (kv.split("=") for kv in argv[i + 1].split(","))}
                i += 2
            else:
                names.append(argv[i])
                i += 1
        res = {}
        for want in names:
            lm = loop_mix(kernel_lines(text, want))
            num = sum(weights.get(k, 0.0) * v[1] for k, v in lm.items())
            den = sum(weights.get(k, 0.0) * v[0] for k, v in lm.items())
            res[want] = dict(loops={k: dict(valu=v[0], fast=v[1]) for k, v in sorted(lm.items())}, passes_per_64_rays=weights,
                             dynamic_fast_fraction=round(num / den, 4) if den else None)
            print("%s: %s -> dynamic fast-opcode fraction %.3f" % (want, ", ".join("%s %d VALU (%d fast)" % (k, v[0], v[1]) for k, v in sorted(lm.items())),
                                                                   num / den if den else 0.0))
        print(json.dumps(res))
    else:
        for want in argv:
            total, fast, ops = static_mix(kernel_lines(text, want))
            print("%s: %d VALU instructions, %d (%.0f %%) in the 2-cycle class" % (want, total, fast, 100.0 * fast / max(total, 1)))
            print("   ", ", ".join("%s %d" % kv for kv in ops.most_common(14)))
