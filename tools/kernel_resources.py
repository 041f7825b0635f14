"""Compile one csrc/*.hip with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs, AGPRs, spills, occupancy (waves/SIMD).  `python tools/kernel_resources.py mrf_fused.hip [filter]`"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "a-modified-hifi-gan-vocoder-using-odconv-and-grc-for-expressive-voice-cloning-_amd", "csrc")
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + extra, cwd=CSRC, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("__hip_bfloat16", "bf16").replace("_Float16", "f16")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|TotalSGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(f"{k:70s} vgpr {v.get('VGPRs', 0):4d} agpr {v.get('AGPRs', 0):4d} spill {v.get('VGPRs Spill', 0):4d} scratch {v.get('ScratchSize [bytes/lane]', 0):5d} occ {v.get('Occupancy [waves/SIMD]', 0)}")
