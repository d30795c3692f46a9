"""Developer aid (here, no GPU): profiles/r04_kernel_resource_usage.txt -- the compiler's kernel-resource-usage remarks for
ecne_engine.hip plus, per device function, the scratch stores / loads counted in the device assembly and its own frame size.
python tools/resource_usage.py > profiles/r04_kernel_resource_usage.txt"""
import os, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
src = os.path.join(ROOT, "ecneproject_amd", "csrc", "ecne_engine.hip")
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src)]
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(base + ["-c", src, "-o", os.path.join(d, "e.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    asm = os.path.join(d, "e.s")
    subprocess.run(base + ["-S", "--cuda-device-only", src, "-o", asm], capture_output=True, text=True, check=True)
    text = open(asm).read()
print("# kernel resource usage, round 6 (hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage on ecneproject_amd/csrc/ecne_engine.hip;")
print("# spill stores / loads per function counted in the device assembly: scratch_store_* / scratch_load_* between the function's label and its .size)")
print()
for ln in r.stderr.splitlines():
    m = re.search(r"remark: (.*)\[-Rpass-analysis", ln)
    if m:
        t = m.group(1).rstrip()
        print(t.strip() if t.lstrip().startswith("Function Name") else "    " + t.strip())
print()
print("%-34s %14s %14s %18s" % ("function", "scratch stores", "scratch loads", "frame bytes (own)"))
rows = []
for m in re.finditer(r"^(_Z\w+):\s*;.*?\n(.*?)^\s*\.size\s+\1,", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    st, ld = len(re.findall(r"\bscratch_store", body)), len(re.findall(r"\bscratch_load", body))
    fm = re.search(r"\.set \.L%s\.private_seg_size, (\d+)" % re.escape(name), text)
    if st + ld >= 4:
        short = re.sub(r"^_ZN4ecne\d*", "", name)[:30]
        rows.append((st + ld, short, st, ld, fm.group(1) if fm else "?"))
for _n, short, st, ld, fr in sorted(rows, reverse=True):
    print("%-34s %14d %14d %18s" % (short, st, ld, fr))
