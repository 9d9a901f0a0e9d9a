"""Register / scratch use of every kernel in a hipcc -S --cuda-device-only listing: python tools/isa_regs.py file.s [name filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)
    v = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)
    acc = re.search(r"\.amdhsa_accum_offset (\d+)", body)
    sc = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)
    rows.append((name, v, acc.group(1) if acc else "-", sc))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (name, v, acc, sc), dn in zip(rows, names):
    dn = dn.replace("mi355ppo::", "").replace("void ", "")
    if flt in dn:
        print(f"vgpr {v:>4} accum_offset {acc:>4} scratch {sc:>5}  {dn[:170]}")
