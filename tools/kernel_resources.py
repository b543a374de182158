#!/usr/bin/env python3
"""Developer tool (build container): registers, spills, occupancy and LDS of the library's kernels, from the compiler's own report.
   python tools/kernel_resources.py [substring ...]      e.g.  k_lossIf k_dual_hIf k_sddmm_mfma"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "sig_sdp_mmw_amd", "csrc", "mmw_api.hip")
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "--cuda-device-only", "-c",
                        "-Rpass-analysis=kernel-resource-usage", src, "-o", os.path.join(d, "x.o")], capture_output=True, text=True)
t = r.stderr
want = sys.argv[1:]
for b in re.split(r"remark: Function Name: ", t)[1:]:
    name = b.split()[0]
    if want and not any(w in name for w in want): continue
    g = lambda k: (re.search(re.escape(k) + r": (\d+)", b) or [None, "?"])[1]
    print("%-70s VGPR %3s AGPR %3s spill %2s scratch %3s waves/SIMD %s LDS %6s" % (name[:70], g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
