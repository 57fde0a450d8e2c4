#!/usr/bin/env python
"""Static check of the gfx950 code objects (no GPU needed): per kernel VGPRs / scratch / LDS, and the two
patterns that cost round 2 time before they were found in the ISA rather than in a counter:
  * dynamic register indexing (s_set_gpr_idx_on / v_movrel*): an accumulator array indexed by a runtime value
    (conv_f32_wino32.hip's epilogue indexed 128 accumulators by the wave's plane half: 72 pairs + 330 v_mov);
  * scratch (private memory) traffic: a local array that was not promoted to registers;
  * (round 4) buffer accesses wrapped in waterfall loops: a descriptor that ended up in VGPRs.

Usage: python tools/isa_lint.py [file.hip ...]        (default: every kernel source under yolo2_light_amd/csrc)
Exit code 1 if any kernel uses scratch, dynamic register indexing or waterfall loops."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yolo2_light_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function",
         "-S", "--cuda-device-only"]


def lint(path: str):
    extra = ["-fno-slp-vectorize"] if path.endswith("conv_f32_wino32.hip") else []
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.check_call([HIPCC] + FLAGS + extra + [path, "-o", tmp.name], stderr=subprocess.DEVNULL, cwd=CSRC)
        text = open(tmp.name).read()
    bad = 0
    # kernels: .amdhsa_kernel <name> ... .end_amdhsa_kernel carry the resource numbers; code sits under "<name>:"
    for m in re.finditer(r"^(\S+):\s*;\s*@\1\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        meta = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(name), text, re.S)
        if not meta:
            continue                      # a device function, not a kernel
        def field(key, default="0"):
            f = re.search(r"\.amdhsa_%s\s+(\S+)" % key, meta.group(1))
            return f.group(1) if f else default
        vgpr = field("next_free_vgpr")
        lds = field("group_segment_fixed_size")
        scratch = int(field("private_segment_fixed_size"))
        dyn = len(re.findall(r"s_set_gpr_idx_on|v_movrel", body))
        # waterfall loop: a buffer instruction whose descriptor the compiler holds in VGPRs (a "uniform" value it computed on
        # the vector unit, e.g. a division): v_readfirstlane x4 + v_cmp_eq_u64 x2 + s_and_saveexec around EVERY access
        # (conv_f32_x3.hip's first version: 11 of them per K-loop iteration)
        wfall = len(re.findall(r"s_and_saveexec_b64[^\n]*\n(?:[^\n]*\n){0,2}?\s*buffer_(?:load|store)", body))
        scr = len(re.findall(r"\bscratch_(load|store)", body))
        flag = ""
        if scratch or scr or dyn or wfall:
            flag = "   <-- " + ", ".join(x for x in ("scratch %d B" % scratch if scratch or scr else "",
                                                   "%d dynamic register index ops" % dyn if dyn else "",
                                                   "%d buffer accesses in waterfall loops" % wfall if wfall else "") if x)
            bad += 1
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        print("%-110s vgpr %4s lds %6s%s" % (short[:110], vgpr, lds, flag))
    return bad


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    bad = 0
    for f in files:
        print("== %s" % os.path.relpath(f, ROOT))
        bad += lint(os.path.abspath(f))
    print("%d kernel(s) flagged" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
