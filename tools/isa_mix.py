"""Static instruction mix of the kernels in a gfx950 assembly file (hipcc -S --cuda-device-only).
    python tools/isa_mix.py /tmp/kr.s [filter]"""
import collections, re, subprocess, sys
lines = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if l.startswith("_ZN4zoic") and ": " in l]
for k, (i, name) in enumerate(starts):
    end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0].replace("void zoic::", "")
    if flt and flt not in short:
        continue
    c = collections.Counter()
    for l in lines[i:end]:
        if not l.startswith("\t"):
            continue
        t = l.strip().split()
        if not t or t[0].startswith((".", ";")):
            continue
        x = t[0]
        if x == "s_endpgm":
            break
        if x.startswith("v_"): c["valu"] += 1
        if x.startswith("s_") and not x.startswith(("s_load", "s_waitcnt", "s_cbranch", "s_branch", "s_nop", "s_setprio")): c["salu"] += 1
        if x.startswith("v_cmp"): c["vcmp"] += 1
        if "readlane" in x or "writelane" in x: c["lane"] += 1
        if x.startswith("s_load"): c["sload"] += 1
        if x.startswith("s_waitcnt"): c["wait"] += 1
        if x.startswith(("s_cbranch", "s_branch")): c["br"] += 1
        if x.startswith(("v_sqrt", "v_rsq", "v_rcp")): c["trans"] += 1
        if x.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
        if x.startswith("ds_"): c["lds"] += 1
        if x.startswith("scratch_"): c["scratch"] += 1
    print("%-34s " % short + " ".join("%s=%d" % kv for kv in sorted(c.items())))
