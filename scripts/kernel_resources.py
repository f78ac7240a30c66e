"""Compact table of hipcc's -Rpass-analysis=kernel-resource-usage remarks for the kernels of csrc/device.hip
(VGPRs, AGPRs, SGPRs, spills, scratch bytes per lane, occupancy).  Usage: python scripts/kernel_resources.py [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "liverrenderer_amd", "csrc", "device.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-fvisibility=hidden",
       "--cuda-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
def demangle(n):
    try: return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception: return n
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
for r in rows:
    print(f"{demangle(r['name'])[:58]:58s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>6s} {r.get('SGPRs Spill','?'):>6s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>7s}")
