"""Compile csrc/gpt_predict.hip with -save-temps and report, per k_var instantiation, register use and whether any
basic block of the hot loop (>= 64 MFMAs, no exp: the reload sweeps) contains scratch (spill) instructions.  A spill
reload there costs more than its own latency: it is a vector-memory operation, so the `s_waitcnt vmcnt` in front of its
use also drains the A-fragment loads in flight (measured round 2: 24 such reloads appeared in the fp64 kernel when the
LDS image was indexed with vector-pointer arithmetic instead of element-pointer arithmetic).
usage: python tools/check_isa_spills.py [-DFLAG ...]   (CPU only; hipcc cross-compiles)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gaussian_process_transportation_amd", "csrc", "gpt_predict.hip")
with tempfile.TemporaryDirectory() as d:
    # device side only (-save-temps also runs the host pass over the intermediate files, which hipcc 7.2 rejects for this source)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-value", "-fno-gpu-rdc",
                    "--cuda-device-only", "-S", SRC, "-o", "dev.s"] + sys.argv[1:], cwd=d, check=True, capture_output=True)
    s = open(os.path.join(d, "dev.s")).read()
bad = 0
for name in re.findall(r"^(_ZN3gpt5k_varI\w+):", s, flags=re.M):
    i = s.index("\n" + name + ":")
    j = s.index(".end_amdhsa_kernel", i)
    body = s[i:j]
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
    blocks, cur, lab = [], [], "entry"
    for line in body.split("\n"):
        if line.startswith(".LBB"):
            blocks.append((lab, cur)); cur = []; lab = line.split(":")[0]
        cur.append(line)
    blocks.append((lab, cur))
    # hot: the ring loops of the diagonal tile (a block that branches to itself, 64 MFMAs) and the sub-chunks of the lock-step part
    # (128 MFMAs each, inside the chunk loop).  The peeled last steps of a diagonal tile (96 / 64 / 32 MFMAs, straight-line, once per
    # tile) are not.
    def is_hot(lab, b):
        n = sum("v_mfma" in x for x in b)
        if any("v_exp" in x or "v_ldexp" in x for x in b):
            return False
        self_loop = any(("s_cbranch" in x or "s_branch" in x) and lab in x for x in b)
        return (n >= 64 and self_loop) or n >= 128
    hot = [b for lab, b in blocks if is_hot(lab, b)]
    spilled = [sum("scratch_" in x for x in b) for b in hot]
    tag = re.search(r"k_varI(\w)Li(\d+)ELb(\d)ELi(\d)ELi(\d+)ELb(\d)ELb(\d)", name).groups()
    total_scratch = sum("scratch_" in x for x in body.split("\n"))
    print(f"k_var<{'double' if tag[0] == 'd' else 'float'}, NCOMP={tag[1]}, CROSS={tag[2]}, KT={tag[3]}, DW={tag[4]}, KSTAR={tag[5]}, HALF={tag[6]}>: vgprs {vg.group(1) if vg else '?'}, "
          f"scratch ops anywhere {total_scratch}, "
          f"{len(hot)} hot blocks, scratch ops in them: {sum(spilled)}")
    if sum(spilled):
        bad += 1
sys.exit(1 if bad else 0)
