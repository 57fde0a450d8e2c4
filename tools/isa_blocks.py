#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -save-temps .s file (offline schedule review).
usage: isa_blocks.py file.s <substring of the mangled kernel name> [--dump LABEL]"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--dump" else None
    s = open(path).read()
    for f in re.split(r"\n(?=_Z\S+:)", s):
        m = re.match(r'(_Z\S+):', f)
        if not m or key not in m.group(1):
            continue
        print(m.group(1))
        blocks, cur = [], None
        for ln in f.split('\n'):
            if re.match(r'\.LBB\d+_\d+:', ln) or cur is None:
                cur = [ln.strip(), []]
                blocks.append(cur)
            elif ln.startswith('\t') and not ln.startswith('\t.') and not ln.startswith('\t;'):
                cur[1].append(ln.strip())
        for lab, ins in blocks:
            c = collections.Counter()
            for i in ins:
                op = i.split()[0]
                for pre, k in (("v_mfma", "mfma"), ("ds_read", "ds_read"), ("ds_load", "ds_read"), ("ds_write", "ds_write"),
                               ("ds_store", "ds_write"), ("buffer_load", "buf_load"), ("buffer_store", "buf_store"),
                               ("global_", "global"), ("scratch_", "SCRATCH"), ("s_waitcnt", "waitcnt"), ("s_barrier", "barrier"),
                               ("s_set_gpr_idx", "GPR_IDX"), ("v_", "valu"), ("s_", "salu")):
                    if op.startswith(pre):
                        c[k] += 1
                        break
                else:
                    c[op] += 1
            print("%-12s %5d %s" % (lab[:12], len(ins), dict(c)))
            if dump and lab.startswith(dump):
                print("\n".join(ins))
        return


main()
