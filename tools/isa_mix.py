"""Static VALU opcode mix of a kernel of librt_hip (hipcc -S): how many of its vector instructions belong
to the class the gfx950 SIMD issues in ~2 cycles per wave64 (v_fma/mul/add/sub_f32, v_mov, v_and/or/xor,
v_add/sub_u32, shifts: measured by tools/issue_microbench.hip) and how many to the ~4-cycle class
(v_min/max/min3/max3, compares, v_cndmask, packed fp32, fp64, 64-bit adds, conversions).
usage: python tools/isa_mix.py <mangled-name substring> [...]"""
import os, re, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = re.compile(r"^v_(fma_f32|fmac_f32|mul_f32|add_f32|sub_f32|subrev_f32|mov_b32|and_b32|or_b32|xor_b32|add_u32|sub_u32|subrev_u32|"
                  r"lshlrev_b32|lshrrev_b32|ashrrev_i32|and_or_b32|lshl_or_b32|lshl_add_u32|add_lshl_u32|or3_b32|bfe_u32|add3_u32)")
out = os.path.join(tempfile.mkdtemp(prefix="isa_"), "rt_hip.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                       "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "raytracing_amd", "csrc"), "--cuda-device-only", "-S",
                       "-o", out, os.path.join(ROOT, "raytracing_amd", "csrc", "rt_hip.hip")], stderr=subprocess.DEVNULL)
text = open(out).read().split("\n")
for want in sys.argv[1:]:
    inside, ops = False, collections.Counter()
    for line in text:
        if re.match(r"^_Z\w*" + re.escape(want) + r"\w*:", line):
            inside = True
            continue
        if inside:
            m = re.match(r"^\s+(v_[a-z0-9_]+)", line)
            if m:
                ops[m.group(1).replace("_e32", "").replace("_e64", "")] += 1
            if line.startswith(".Lfunc_end"):
                break
    total = sum(ops.values())
    fast = sum(n for o, n in ops.items() if FAST.match(o))
    print("%s: %d VALU instructions, %d (%.0f %%) in the 2-cycle class" % (want, total, fast, 100.0 * fast / max(total, 1)))
    print("   ", ", ".join("%s %d" % kv for kv in ops.most_common(14)))
