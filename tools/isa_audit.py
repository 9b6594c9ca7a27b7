"""Per-kernel code-generation audit of the HIP sources: instructions, branches, loads that are waited on with vmcnt(0) right behind
them (serialised loads), readfirstlane waterfall loops, VGPRs and scratch.  python tools/isa_audit.py [file.hip ...]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "tatt_amd/csrc/*.hip")))
print(f"{'kernel':64s} {'instr':>6s} {'br':>4s} {'ld':>4s} {'ser':>4s} {'wfall':>5s} {'vgpr':>5s} {'scr':>5s}")
for f in src:
    with tempfile.NamedTemporaryFile(suffix=".s") as t:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", t.name, f,
                        "-I" + os.path.join(root, "tatt_amd/csrc")], check=True, stderr=subprocess.DEVNULL)
        lines = open(t.name).read().split("\n")
    cur, st, meta = None, {}, {}
    prev_load = False
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            st[cur] = dict(instr=0, br=0, ld=0, ser=0, wf=0)
            continue
        m = re.match(r"\s+\.amdhsa_kernel (\S+)", l)
        if m:
            cur = None
        if cur is None:
            continue
        t_ = l.strip()
        if not t_ or t_.startswith(";") or t_.startswith("."):
            continue
        op = t_.split()[0]
        s = st[cur]
        s["instr"] += 1
        if op.startswith("s_cbranch"):
            s["br"] += 1
        if op.startswith(("global_load", "buffer_load", "flat_load")):
            s["ld"] += 1
            prev_load = True
            continue
        if op == "v_readfirstlane_b32":
            s["wf"] += 1
        if prev_load and op == "s_waitcnt" and "vmcnt(0)" in t_:
            s["ser"] += 1
        prev_load = False
        if op == "s_endpgm":
            cur = None
    txt = "\n".join(lines)
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size: (\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", txt):
        meta[m.group(1)] = (int(m.group(3)), int(m.group(2)))
    for k, s in st.items():
        v, sc = meta.get(k, (0, 0))
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        name = name[:name.index("(")] if "(" in name else name
        print(f"{os.path.basename(f)[:12]:12s} {name[:51]:51s} {s['instr']:6d} {s['br']:4d} {s['ld']:4d} {s['ser']:4d} {s['wf']:5d} {v:5d} {sc:5d}")
